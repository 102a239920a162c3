#!/bin/bash
# timing-only ablations of the dataflow kernel at 4K (results wrong): where do its ~28 us per item go?
for v in default; do
  if [ $v = default ]; then unset RTDD_LIBRARY; else export RTDD_LIBRARY=$PWD/realtimedepthdiffusion_amd/librtdd_$v.so; fi
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-estimate --workload 4k_jacobi1000 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v mode', d['config']['persistent'], 'Gpx-it/s %.1f' % (d['value']/1e3), 'ms %.3f' % d['ms_per_step'])"
done
