#!/bin/bash
# lane shifts through the LDS crossbar instead of DPP: parity of the blocked kernels, then throughput (default library, then A/B variants named in $VARIANTS)
set -o pipefail
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3_bperm_tests.txt 2>&1 || { tail -40 gpurun_out/r3_bperm_tests.txt; exit 1; }
tail -2 gpurun_out/r3_bperm_tests.txt
fi
one() {  # workload tile depth persistent
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-estimate --workload $1 --tile $2 --temporal-depth $3 --persistent $4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); it=int('$1'.split('x')[2]) if 'x' in '$1' and '$1'[0].isdigit() else 1000
print('${RTDD_LIBRARY##*/} $1 tile $2 depth $3 persistent-option $4 ->', d['config']['tile'], d['config']['temporal_depth'], 'mode', d['config']['persistent'], 'Gpx-it/s %.1f' % (d['value']/1e3), 'ms %.3f' % d['ms_per_step'], 'us/sweep %.4f' % (d['ms_per_step']*1e3/it))"
}
for v in default $VARIANTS; do
  if [ $v = default ]; then unset RTDD_LIBRARY; else export RTDD_LIBRARY=$PWD/realtimedepthdiffusion_amd/librtdd_$v.so; fi
  one 96x128x4000 4 8 0
  one 1080p_jacobi1000 0 0 1
  one 4k_jacobi1000 0 0 1
  one 8k_jacobi200 0 0 1
  python3 scripts/estimate_bench.py 2>/dev/null | tail -3
done
