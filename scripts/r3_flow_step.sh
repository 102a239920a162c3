#!/bin/bash
set -o pipefail
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dataflow or randomised_shapes" > gpurun_out/r3_flow_tests.txt 2>&1 || { tail -30 gpurun_out/r3_flow_tests.txt; exit 1; }
tail -2 gpurun_out/r3_flow_tests.txt
timeout -k 10 500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "north_star or large_tile or one_launch_per_block" > gpurun_out/r3_flow_tests2.txt 2>&1 || { tail -30 gpurun_out/r3_flow_tests2.txt; exit 1; }
tail -2 gpurun_out/r3_flow_tests2.txt
for wl in 4k_jacobi1000 8k_jacobi200; do for pz in 1 0; do
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-estimate --workload $wl --persistent $pz 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl persistent-option $pz ->', d['config']['tile'], d['config']['temporal_depth'], 'mode', d['config']['persistent'], 'Gpx-it/s %.1f' % (d['value']/1e3), 'ms %.3f' % d['ms_per_step'])"
done; done
