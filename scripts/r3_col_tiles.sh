#!/bin/bash
# column-layout tiles of 16 / 12 / 8 waves on the two coarsest levels of the 1080p cascade
for wl in 120x67_jacobi1000 240x135_jacobi500; do
  echo "== $wl"
  for td in "0 0" "14 28" "14 24" "16 20" "16 16" "16 12" "15 12" "15 10" "15 8"; do set -- $td
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-estimate --workload $wl --tile $1 --temporal-depth $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile $1 depth $2 ->', d['config']['tile'], d['config']['temporal_depth'], 'ms %.4f' % d['ms_per_step'])"
  done
done
