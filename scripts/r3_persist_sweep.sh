#!/bin/bash
# tile x depth table of the persistent mode after round 3's cheaper hand-off, for the three finest levels of the 1080p cascade
for wl in 480x270_jacobi250 960x540_jacobi125 1080x1920x62; do
  echo "== $wl auto:"; python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-estimate --workload $wl 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('auto tile', d['config']['tile'], 'depth', d['config']['temporal_depth'], 'persistent', d['config']['persistent'], 'ms %.4f' % d['ms_per_step'])"
  for tile in 4 9 8 5 6 7 1 2; do for depth in 4 6 8 12 16; do
    python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-estimate --workload $wl --tile $tile --temporal-depth $depth --persistent 1 2>/dev/null | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile $tile depth $depth ->', 'persistent', d['config']['persistent'], 'ms %.4f' % d['ms_per_step'])
except Exception as e: print('tile $tile depth $depth failed')"
  done; done
done
