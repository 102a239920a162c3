#!/bin/bash
# A round's profiles, part 2: one estimate (timeline + summary of the SAME estimate), pipelined live frames (timeline of one frame), the defocus
# pipeline per kernel, BASELINE configs[3] at N = 1 as solves and as batched estimates, the default line.   usage: ROUND=r05 bash scripts/profile_extra.sh
set -o pipefail
R=$GRAFT_REPO_ROOT; RD=${ROUND:-r05}; P=$R/gpurun_out/profiles_$RD; mkdir -p $P
OUT=$R/gpurun_out/prof_estimate_$RD; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o est -- python3 $R/scripts/prof_estimate.py > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
rm -rf /tmp/lv && rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/lv -o lv --output-format csv -- python3 $R/scripts/prof_live.py > /dev/null 2>&1 || exit 1
rm -rf /tmp/df && rocprofv3 --kernel-trace --stats -d /tmp/df -o df --output-format csv -- python3 $R/scripts/prof_defocus.py > /dev/null 2>&1 || exit 1
cd $R
python3 scripts/prof_estimate_report.py $OUT $P/${RD}_estimate || exit 1
python3 scripts/prof_live_report.py /tmp/lv > $P/${RD}_live_frame_timeline.txt || exit 1
(cd /tmp && rm -rf /tmp/lvfx && rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/lvfx -o lv --output-format csv -- python3 $R/scripts/prof_live.py defocus > /dev/null 2>&1) || exit 1
python3 scripts/prof_live_report.py /tmp/lvfx > $P/${RD}_live_frame_defocus_timeline.txt || exit 1
find /tmp/df -name '*kernel_stats.csv' | head -1 | xargs -I{} sh -c "cut -c1-200 {} | head -12" > $P/${RD}_defocus_kernel_stats.csv
python3 bench.py --gpus 1 --workload batch64_1080p --steps 2 --warmup 1 --no-cpu-baseline --verify > $P/${RD}_batch64_1080p_n1.json 2>/dev/null || exit 1
python3 bench.py --gpus 1 --workload batch64_1080p_estimate --steps 3 --warmup 1 --no-cpu-baseline --verify > $P/${RD}_batch64_1080p_estimate_n1.json 2>/dev/null || exit 1
python3 bench.py > $P/${RD}_bench_default.json 2>/dev/null || exit 1
tail -c 400 $P/${RD}_bench_default.json
