#!/bin/bash
# the whole GPU suite, then the driver's bench line
set -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3_full_tests_3.txt 2>&1 || { tail -40 gpurun_out/r3_full_tests_3.txt; exit 1; }
tail -3 gpurun_out/r3_full_tests_3.txt
python3 bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err || { tail -20 gpurun_out/r3_bench_default.err; exit 1; }
python3 -c "
import json; d = json.loads(open('gpurun_out/r3_bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], d['unit'], 'ms', d['ms_per_step'], 'roofline', d['roofline'])
for k in ('estimate', 'estimate_4k', 'sweep_4k', 'effects', 'cpu_baseline'): print(k, d.get(k))"
