#!/bin/bash
one() {  # workload tile depth persistent
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-estimate --workload $1 --tile $2 --temporal-depth $3 --persistent $4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); it=int('$1'.split('x')[2]) if 'x' in '$1' and '$1'[0].isdigit() else 1000
print('${RTDD_LIBRARY##*/} $1 tile $2 depth $3 persistent-option $4 ->', d['config']['tile'], d['config']['temporal_depth'], 'mode', d['config']['persistent'], 'Gpx-it/s %.1f' % (d['value']/1e3), 'ms %.3f' % d['ms_per_step'], 'us/sweep %.4f' % (d['ms_per_step']*1e3/it))"
}
for v in $VARIANTS; do
  if [ $v = default ]; then unset RTDD_LIBRARY; else export RTDD_LIBRARY=$PWD/realtimedepthdiffusion_amd/librtdd_$v.so; fi
  for t in $TILES; do one 96x128x4000 $t 8 0; done
done
