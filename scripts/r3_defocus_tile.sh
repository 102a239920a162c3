#!/bin/bash
# the one-launch defocus (per-tile summed-area tables in LDS): parity of both paths, then wall times by path
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_effects_fullsize.py -x -q -m gpu -k "defocus or effects" > gpurun_out/r3_dtile_tests.txt 2>&1 || { tail -40 gpurun_out/r3_dtile_tests.txt; exit 1; }
tail -2 gpurun_out/r3_dtile_tests.txt
python3 scripts/defocus_paths.py 2>/dev/null
