#!/bin/bash
# defocus rewrite: parity tests, then per-kernel times + counters
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_effects_fullsize.py tests/test_gpu_dataset.py tests/test_golden_gpu.py -x -q -m gpu > gpurun_out/r3_defocus_tests.txt 2>&1 || { tail -40 gpurun_out/r3_defocus_tests.txt; exit 1; }
tail -3 gpurun_out/r3_defocus_tests.txt
bash scripts/r3_defocus_prof.sh
python3 scripts/effects_bench.py 2>&1 | grep -i "defocus\|---"
