# Round profile: kernel-trace stats + HBM traffic counters (separate --pmc passes, as MI355X_MICROARCH.md prescribes)
export TMPDIR=/tmp
R=${ROUND:-r01}; WL=${WL:-1080p_jacobi1000}
OUT=gpurun_out/prof_$R_$WL; mkdir -p $OUT
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-estimate --workload $WL"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python3 $ARGS > $OUT/write.log 2>&1
python3 bench.py --steps 10 --warmup 2 --workload $WL > $OUT/bench.json 2>$OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-600
