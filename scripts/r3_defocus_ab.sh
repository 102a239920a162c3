#!/bin/bash
# band-height A/B of the defocus table build (kernel-trace per-kernel averages)
set -o pipefail
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_effects_fullsize.py -x -q -m gpu > gpurun_out/r3_defocus_tests.txt 2>&1 || { tail -40 gpurun_out/r3_defocus_tests.txt; exit 1; }
tail -2 gpurun_out/r3_defocus_tests.txt
cd /tmp && export TMPDIR=/tmp
for rb in 0 4 8 16 32; do
  OUT=$R/gpurun_out/prof_defocus_rb$rb; rm -rf $OUT; mkdir -p $OUT
  RTDD_DEFOCUS_BAND=$rb rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o df -- python3 $R/scripts/prof_defocus.py > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
  echo "== band $rb"; python3 $R/scripts/prof_defocus_summary.py $OUT
done
