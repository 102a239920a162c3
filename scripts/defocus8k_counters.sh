#!/bin/bash
# 8K defocus (smooth depth map, 10 calls each): kernel time and HBM-side bytes (FETCH_SIZE / WRITE_SIZE, each --pmc group in a pass of its
# own) for three forms of the table path -- one table / row bands per XCD (round 5), one table / column strips per XCD (round 6 default),
# five slices of <= 64 MB / column strips (the banded table VERDICT r5 asked for) -> gpurun_out/r06_defocus_counters.json
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_defocus8k_counters; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "rowbands 1 0" "strips 2 0" "slices64 2 64"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1 -o t -- python3 $R/scripts/prof_defocus8k.py $2 $3 > $OUT/$1.trace.log 2>&1 || { tail -5 $OUT/$1.trace.log; exit 1; }
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$1 -o f -- python3 $R/scripts/prof_defocus8k.py $2 $3 > $OUT/$1.fetch.log 2>&1 || { tail -5 $OUT/$1.fetch.log; exit 1; }
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$1 -o w -- python3 $R/scripts/prof_defocus8k.py $2 $3 > $OUT/$1.write.log 2>&1 || { tail -5 $OUT/$1.write.log; exit 1; }
done
cd $R && python3 scripts/defocus8k_counters_report.py $OUT > gpurun_out/r06_defocus_counters.json && cat gpurun_out/r06_defocus_counters.json | head -60
