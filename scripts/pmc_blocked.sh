# per-kernel PMC pass for the blocked sweep kernel (counters only; no trace domains combined with --pmc besides kernel-trace)
export TMPDIR=/tmp
OUT=gpurun_out/pmc1; mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --sweep-kernel 2 --tile ${TILE:-1} --temporal-depth ${DEPTH:-8} --workload ${WL:-1080p_jacobi1000}"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o sq -- python3 $ARGS > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT -o sq2 -- python3 $ARGS > $OUT/sq2.log 2>&1
ls $OUT
python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc1/*counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:40]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    for k,v in acc.items():
        if 'sweep' in k: print(f.split('/')[-1], k, {a: round(b) for a,b in v.items()})
PY
