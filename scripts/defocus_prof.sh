#!/bin/bash
# per-kernel times of the defocus pipeline (scripts/prof_defocus.py: 20 calls x {1080p, 4K} x {random depth, smooth depth}) + FETCH_SIZE / WRITE_SIZE passes
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_defocus
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o df -- python3 $R/scripts/prof_defocus.py > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python3 $R/scripts/prof_defocus.py > $OUT/fetch.log 2>&1 || { tail -20 $OUT/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python3 $R/scripts/prof_defocus.py > $OUT/write.log 2>&1 || { tail -20 $OUT/write.log; exit 1; }
cd $R && python3 scripts/prof_defocus_report.py $OUT
