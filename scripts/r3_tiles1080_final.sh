#!/bin/bash
# 1080p persistent: the tiles that hold 128 x 96 (16 / 12 / 8 waves) with the final kernels
for t in "0 0" "4 8" "12 8" "13 8" "3 8" "8 8"; do set -- $t
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-estimate --workload 1080p_jacobi1000 --tile $1 --temporal-depth $2 --persistent 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile $1 depth $2 ->', d['config']['tile'], d['config']['temporal_depth'], 'mode', d['config']['persistent'], 'Gpx-it/s %.1f' % (d['value']/1e3), 'ms %.3f' % d['ms_per_step'])"
done
