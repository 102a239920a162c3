#!/bin/bash
# where does a sweep of the packed kernel go?  (1) ONE tile, all sweeps in one launch, no halo: us per sweep by tile; (2) timing-only ablations
one() {  # workload tile depth persistent
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-estimate --workload $1 --tile $2 --temporal-depth $3 --persistent $4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); it=int('$1'.split('x')[2]) if 'x' in '$1' and '$1'[0].isdigit() else 1000
print('${RTDD_LIBRARY##*/} $1 tile $2 depth $3 persistent-option $4 ->', d['config']['tile'], d['config']['temporal_depth'], 'mode', d['config']['persistent'], 'Gpx-it/s %.1f' % (d['value']/1e3), 'ms %.3f' % d['ms_per_step'], 'us/sweep %.4f' % (d['ms_per_step']*1e3/it))"
}
unset RTDD_LIBRARY
for t in 4 17 18; do one 96x128x4000 $t 8 0; done
for t in 6 19; do one 96x64x4000 $t 8 0; done
one 48x128x4000 17 8 0; one 48x128x4000 4 8 0
for v in nolds nopoll notiny relaxed; do
  export RTDD_LIBRARY=$PWD/realtimedepthdiffusion_amd/librtdd_pk_$v.so
  one 96x128x4000 17 8 0; one 96x128x4000 18 8 0
  one 1080p_jacobi1000 17 8 1; one 1080p_jacobi1000 18 8 1; one 4k_jacobi1000 17 8 0
done
