#!/bin/bash
# Round-3 profiles beyond the four bench workloads: the estimate's kernel trace, the defocus pipeline (trace + FETCH / WRITE passes),
# the N = 1 base of BASELINE configs[3], and the default bench line.  Summaries -> gpurun_out/profiles_r03/ (copied into profiles/).
set -o pipefail
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/profiles_r03; mkdir -p $P
cd $R
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r3_kernels_tests.txt 2>&1 || { tail -30 gpurun_out/r3_kernels_tests.txt; exit 1; }
tail -1 gpurun_out/r3_kernels_tests.txt
bash scripts/r3_estimate_prof.sh > $P/r03_estimate_timeline.txt 2>&1 || { tail $P/r03_estimate_timeline.txt; exit 1; }
python3 scripts/prof_estimate_csv.py gpurun_out/prof_estimate_r3 > $P/r03_estimate_kernel_trace_summary.csv
bash scripts/r3_defocus_prof.sh > $P/r03_defocus_kernels.txt 2>&1 || { tail $P/r03_defocus_kernels.txt; exit 1; }
cp gpurun_out/prof_defocus_r3/df_kernel_stats.csv $P/r03_defocus_1080p_4k_kernel_stats.csv
python3 scripts/prof_defocus_report.py gpurun_out/prof_defocus_r3 --json > $P/r03_defocus_counters.json
python3 bench.py --gpus 1 --workload batch64_1080p --steps 2 --warmup 1 --no-cpu-baseline --verify > $P/r03_batch64_1080p_n1.json 2>/dev/null || exit 1
python3 bench.py > $P/r03_bench_default.json 2>/dev/null || exit 1
tail -c 600 $P/r03_bench_default.json
