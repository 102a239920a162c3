#!/bin/bash
# SQ counters of the "no Dirichlet select" sweep variant (EXPERIMENTS.md, round 4) against the build without it: one --pmc pass each.
export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
for wl in 1080p_jacobi1000 4k_jacobi1000; do for v in nodir0 nodir1; do
  OUT=gpurun_out/prof_r04_nodir/${wl}_$v; mkdir -p $OUT
  RTDD_LIBRARY=$PWD/realtimedepthdiffusion_amd/librtdd_$v.so rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $OUT -o sq1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-estimate --no-clock-ramp --workload $wl > $OUT/sq1.log 2>&1 || { tail $OUT/sq1.log; exit 1; }
  python3 - $OUT $wl $v <<'PY'
import csv, sys, collections
d, wl, v = sys.argv[1:4]
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for r in csv.DictReader(open(d + "/sq1_counter_collection.csv")):
    if "k_sweep_blocked" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
per = {c: acc[c] / len(n[c]) for c in acc}
print(wl, v, "per launch:", " ".join(f"{c}={per[c]:.4g}" for c in sorted(per)), "| VALU active/wave cycles %.3f wait_any/wave cycles %.3f" % (per["SQ_ACTIVE_INST_VALU"] / per["SQ_WAVE_CYCLES"], per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"]))
PY
done; done
