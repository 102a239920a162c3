#!/bin/bash
# the column-layout kernel (coarse pyramid levels) with its lane shifts through the LDS crossbar: parity of that kernel, then the estimate
set -o pipefail
export RTDD_LIBRARY=$PWD/realtimedepthdiffusion_amd/librtdd_colbp.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "column or 14- or randomised or solver_bit_exact" > gpurun_out/r3_colbp_tests.txt 2>&1 || { tail -30 gpurun_out/r3_colbp_tests.txt; exit 1; }
tail -2 gpurun_out/r3_colbp_tests.txt
python3 scripts/estimate_bench.py 2>/dev/null | tail -6
unset RTDD_LIBRARY
python3 scripts/estimate_bench.py 2>/dev/null | tail -6
