#!/bin/bash
# Round-4 profiles, part 2: one estimate (timeline + summary of the SAME estimate), the defocus pipeline, configs[3] at N = 1, the default line.
set -o pipefail
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/profiles_r04; mkdir -p $P
OUT=$R/gpurun_out/prof_estimate_r4; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o est -- python3 $R/scripts/prof_estimate.py > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
cd $R
python3 scripts/prof_estimate_report.py $OUT $P/r04_estimate || exit 1
python3 bench.py --gpus 1 --workload batch64_1080p --steps 2 --warmup 1 --no-cpu-baseline --verify > $P/r04_batch64_1080p_n1.json 2>/dev/null || exit 1
python3 bench.py > $P/r04_bench_default.json 2>/dev/null || exit 1
tail -c 400 $P/r04_bench_default.json
