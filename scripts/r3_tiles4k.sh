#!/bin/bash
# tile x depth at 4K and 8K (launch per block) after round 3's changes
for wl in 4k_jacobi1000 8k_jacobi200; do echo "== $wl"
for t in "0 0" "6 8" "6 12" "4 8" "4 12" "4 16" "5 8" "7 8" "8 8" "10 8" "12 8" "12 12"; do set -- $t
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-estimate --workload $wl --tile $1 --temporal-depth $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile $1 depth $2 ->', d['config']['tile'], d['config']['temporal_depth'], 'Gpx-it/s %.1f' % (d['value']/1e3), 'launch_us %.1f' % d['roofline']['launch_us'])"
done; done
