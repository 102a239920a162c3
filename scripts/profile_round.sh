# Round profile of ONE bench workload: a clean kernel-trace pass at the driver's --steps 20 --warmup 5, then counters in passes of
# their own (as MI355X_MICROARCH.md prescribes: --pmc only ever beside --kernel-trace): FETCH_SIZE, WRITE_SIZE, two SQ passes.
# usage: ROUND=r02 WL=1080p_jacobi1000 bash scripts/profile_round.sh     -> gpurun_out/prof_${ROUND}_$WL/ ; then scripts/make_counters_json.py
export TMPDIR=/tmp
R=${ROUND:-r05}; WL=${WL:-1080p_jacobi1000}
OUT=gpurun_out/prof_${R}_$WL; mkdir -p $OUT
FULL="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-estimate --workload $WL $EXTRA"
SHORT="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-estimate --workload $WL $EXTRA"
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM"
python3 $FULL > $OUT/bench_unprofiled.json 2>$OUT/bench.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $FULL > $OUT/trace.json 2>$OUT/trace.err &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python3 $SHORT > $OUT/fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python3 $SHORT > $OUT/write.log 2>&1 &&
rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $OUT -o sq1 -- python3 $SHORT > $OUT/sq1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $OUT -o sq2 -- python3 $SHORT > $OUT/sq2.log 2>&1 &&
echo "profiled $WL -> $OUT"
