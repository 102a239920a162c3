#!/bin/bash
# the two coarsest levels of the 1080p cascade by temporal depth of the column-layout kernel (final kernels)
one() {
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-estimate --workload $1 --tile $2 --temporal-depth $3 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 tile $2 depth $3 ->', d['config']['tile'], 'ms %.4f' % d['ms_per_step'])"
}
one 67x120x1000 0 0; for T in 16 20 24 26 28; do one 67x120x1000 14 $T; done
one 135x240x500 0 0; for T in 16 20 24 26 28; do one 135x240x500 14 $T; done
one 270x480x250 0 0; for T in 12 16 20 24; do one 270x480x250 14 $T; done
