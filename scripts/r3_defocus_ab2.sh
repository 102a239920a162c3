#!/bin/bash
# rows-per-wave A/B of the defocus lookup (kernel-trace per-kernel averages)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in rows1 rows2 default rows8; do
  OUT=$R/gpurun_out/prof_defocus_$v; rm -rf $OUT; mkdir -p $OUT
  if [ $v = default ]; then unset RTDD_LIBRARY; else export RTDD_LIBRARY=$R/realtimedepthdiffusion_amd/librtdd_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o df -- python3 $R/scripts/prof_defocus.py > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
  echo "== $v"; python3 $R/scripts/prof_defocus_summary.py $OUT | grep "k_defocus\|total"
done
