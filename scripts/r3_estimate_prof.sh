#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_estimate_r3
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RTDD_DEBUG_CONFIG=1 rocprofv3 --kernel-trace --output-format csv -d $OUT -o est -- python3 $R/scripts/prof_estimate.py > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
cd $R && grep "rtdd\]" $OUT/trace.log | sort | uniq -c | head -20; python3 scripts/prof_estimate_timeline.py $OUT; python3 scripts/prof_estimate_summary.py $OUT
