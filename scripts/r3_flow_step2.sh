#!/bin/bash
for w in 0 2 3 4; do
RTDD_DEBUG_CONFIG=1 RTDD_FLOW_WGS=$w python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-estimate --workload 4k_jacobi1000 2>gpurun_out/flow_dbg_$w.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgs/CU $w ->', d['config']['tile'], 'mode', d['config']['persistent'], 'Gpx-it/s %.1f' % (d['value']/1e3), 'ms %.3f' % d['ms_per_step'])"
grep dataflow gpurun_out/flow_dbg_$w.txt | head -1
done
