#!/bin/bash
# is the automatic (tile, depth, persistence) still the best of a few fixed candidates?  assorted sizes x 400 sweeps
for sz in 200x200 240x320 512x512 480x640 600x800 720x1280 1080x1920; do
  for cfg in "0 0 -1" "14 28 0" "14 16 0" "9 16 1" "9 8 1" "4 8 1" "6 8 0" "5 8 1"; do set -- $cfg
    v=$(RTDD_DEBUG_CONFIG=1 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-estimate --workload ${sz}x400 --tile $1 --temporal-depth $2 --persistent $3 2>/tmp/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f Gpx-it/s %.4f ms' % (d['value']/1e3, d['ms_per_step']))")
    echo "$sz tile $1 depth $2 persistent $3: $v  $(grep -m1 rtdd /tmp/err.txt | cut -c1-70)"
  done
done
