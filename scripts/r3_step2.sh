#!/bin/bash
set -o pipefail
python -m pytest tests/test_gpu_cascade.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_golden_gpu.py -x -q -m gpu > gpurun_out/r3_step2_tests.txt 2>&1 || { tail -30 gpurun_out/r3_step2_tests.txt; exit 1; }
tail -2 gpurun_out/r3_step2_tests.txt
for i in 1 2; do python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('estimate', d['estimate']['ms'], 'estimate_4k', d['estimate_4k']['ms'], 'value', d['value'], 'sweep_4k', d['sweep_4k']['value'])"; done
