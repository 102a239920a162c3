#!/bin/bash
# Round-3 profiles: the four bench workloads (kernel trace + FETCH / WRITE / two SQ passes each), then the summaries under profiles/
set -o pipefail
cd $GRAFT_REPO_ROOT
for wl in 1080p_jacobi1000 4k_jacobi1000 4k_rbsor_1e-4 8k_multigrid_1e-4; do
  ROUND=r03 WL=$wl bash scripts/profile_round.sh || { echo "profile of $wl failed"; tail -5 gpurun_out/prof_r03_$wl/*.err gpurun_out/prof_r03_$wl/*.log | tail -30; exit 1; }
  python3 scripts/make_counters_json.py r03 $wl || exit 1
done
mkdir -p gpurun_out/profiles_r03 && cp profiles/r03_* profiles/counters_latest.json gpurun_out/profiles_r03/
