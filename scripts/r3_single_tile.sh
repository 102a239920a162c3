#!/bin/bash
# per-sweep time of ONE tile (all sweeps in one launch, no halo, every wave stays): the latency chain of a sweep by wave count
for wl in 4x64x4000 8x64x4000 16x64x4000 32x64x4000 64x64x4000; do
  for t in 14 9; do
    python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-estimate --workload $wl --tile $t --temporal-depth 8 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); it=int('$wl'.split('x')[2]); print('$wl tile $t ->', d['config']['tile'], 'launches/solve sweeps_per_launch', d['config']['sweeps_per_launch'], 'us/sweep %.4f' % (d['ms_per_step']*1e3/it), 'launch_us', round(d['roofline']['launch_us'],1))"
  done
done
