#!/bin/bash
# the packed tiles (sweep_pk.hip): parity first, then throughput beside the scalar tiles in the same call
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed or (blocked_kernel and (17- or 18- or 19-)) or (persistent_mode and (17- or 18- or 19-))" > gpurun_out/r3_pk_tests.txt 2>&1 || { tail -40 gpurun_out/r3_pk_tests.txt; exit 1; }
tail -2 gpurun_out/r3_pk_tests.txt
run() {  # workload tile depth persistent
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-estimate --workload $1 --tile $2 --temporal-depth $3 --persistent $4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 tile $2 depth $3 persistent-option $4 ->', d['config']['tile'], d['config']['temporal_depth'], 'mode', d['config']['persistent'], 'Gpx-it/s %.1f' % (d['value']/1e3), 'ms %.3f' % d['ms_per_step'])"
}
for t in 4 17 18; do run 1080p_jacobi1000 $t 8 1; done
for t in 4 17 18; do run 1080p_jacobi1000 $t 8 0; done
for t in 6 19 17 18; do run 4k_jacobi1000 $t 8 0; done
run 4k_jacobi1000 19 12 0
run 8k_jacobi200 6 8 0; run 8k_jacobi200 19 8 0
