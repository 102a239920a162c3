#!/bin/bash
# scripts/r3_ab_generic.sh "variants" : persistent parity tests on the default build, then A/B of the listed library variants
set -o pipefail
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "persistent or full_size or randomised or status or rbgs" > gpurun_out/r3_ab_tests.txt 2>&1 || { tail -30 gpurun_out/r3_ab_tests.txt; exit 1; }
tail -2 gpurun_out/r3_ab_tests.txt
for i in 1 2; do bash scripts/ab_variants.sh "$1" "1080p_jacobi1000 960x540_jacobi125 480x270_jacobi250" --no-estimate; done
