# Round profile: kernel-trace stats + HBM traffic counters (separate --pmc passes, as MI355X_MICROARCH.md prescribes)
export TMPDIR=/tmp
R=${ROUND:-r01}; WL=${WL:-1080p_jacobi1000}
OUT=gpurun_out/prof_$R_$WL; mkdir -p $OUT
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $WL"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python3 $ARGS > $OUT/write.log 2>&1
python3 bench.py --steps 10 --warmup 2 --workload $WL > $OUT/bench.json 2>$OUT/bench.err
python3 - "$OUT" <<'PY'
import csv, collections, sys, json
out=sys.argv[1]
print(open(f'{out}/trace_kernel_stats.csv').read()[:1500])
res={}
for name,cn in (('fetch','FETCH_SIZE'),('write','WRITE_SIZE')):
    acc=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f'{out}/{name}_counter_collection.csv')):
        if r['Counter_Name']==cn:
            k=r['Kernel_Name'].split('(')[0][-60:]; acc[k]+=float(r['Counter_Value']); n[k]+=1
    for k in acc: res.setdefault(k,{})[cn]=(acc[k]/n[k], n[k])
for k,v in res.items(): print(k, {a:(round(b[0],1), b[1]) for a,b in v.items()})
print(open(f'{out}/bench.json').read())
PY
