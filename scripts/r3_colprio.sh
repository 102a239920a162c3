#!/bin/bash
# wave priority (s_setprio) around the rows the neighbouring waves wait for: the column-layout kernel (default since round 3) and, as variants, raised from the top of the sweep / the same in k_sweep_blocked
one() {
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-estimate --workload $1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${RTDD_LIBRARY##*/} $1 ->', d['config']['tile'], d['config']['temporal_depth'], 'mode', d['config']['persistent'], 'Gpx-it/s %.1f' % (d['value']/1e3), 'ms %.3f' % d['ms_per_step'])"
}
for v in default $VARIANTS; do
  if [ $v = default ]; then unset RTDD_LIBRARY; else export RTDD_LIBRARY=$PWD/realtimedepthdiffusion_amd/librtdd_$v.so; fi
  echo "== $v"; python3 scripts/estimate_bench.py 2>/dev/null | head -6
  one 1080p_jacobi1000; one 4k_jacobi1000
done
