#!/bin/bash
# per-level timings of the 1080p cascade's solves under a few (tile, depth, persistent) choices: scripts/small_levels.sh > gpurun_out/small_levels.txt
for wl in 120x67_jacobi1000 240x135_jacobi500 480x270_jacobi250 960x540_jacobi125; do
  for cfg in "0 0 -1" "9 24 0" "9 24 1" "9 16 1" "9 8 1" "6 8 1" "6 16 1" "7 8 1" "5 8 1" "4 8 1" "4 8 0" "4 24 0"; do set -- $cfg
    v=$(RTDD_DEBUG_CONFIG=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-estimate --workload $wl --tile $1 --temporal-depth $2 --persistent $3 2>/tmp/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f Gpx-it/s %.4f ms' % (d['value']/1e3, d['ms_per_step']))")
    echo "$wl tile $1 depth $2 persistent $3: $v  $(grep -m1 rtdd /tmp/err.txt)"
  done
done
