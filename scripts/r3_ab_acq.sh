#!/bin/bash
set -o pipefail
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent or full_size or randomised or status" > gpurun_out/r3_acq_tests.txt 2>&1 || { tail -30 gpurun_out/r3_acq_tests.txt; exit 1; }
tail -2 gpurun_out/r3_acq_tests.txt
for i in 1 2; do bash scripts/ab_variants.sh "acq default" "1080p_jacobi1000 960x540_jacobi125 480x270_jacobi250" --no-estimate; done
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('estimate', d['estimate']['ms'], 'estimate_4k', d['estimate_4k']['ms'], 'value', d['value'])"
