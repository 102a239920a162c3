#!/bin/bash
# A/B of library variants (scripts/build_variant.sh): scripts/ab_variants.sh "v0 v1 default" "1080p_jacobi1000 4k_jacobi1000" [extra bench args]
# prints one line per (variant, workload): value in Gpx-it/s, ms per step, estimate ms
VARS=$1; WLS=$2; shift 2
for wl in $WLS; do for v in $VARS; do
    if [ $v = default ]; then unset RTDD_LIBRARY; else export RTDD_LIBRARY=$PWD/realtimedepthdiffusion_amd/librtdd_$v.so; fi
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $wl "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', '$wl', 'Gpx-it/s %.1f' % (d['value'] / 1e3), 'ms/step %.4f' % d['ms_per_step'], 'estimate_ms', d.get('estimate', {}).get('ms'))
"
done; done
