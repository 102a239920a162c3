#!/bin/bash
# Round-4 profiles, part 1: the bench workloads (kernel trace + FETCH / WRITE / two SQ passes each), then the summaries under profiles/.
# usage (on the GPU box): bash scripts/r4_profile_all.sh "1080p_jacobi1000 4k_jacobi1000 8k_jacobi200"
set -o pipefail
cd $GRAFT_REPO_ROOT
for wl in ${1:-1080p_jacobi1000 4k_jacobi1000 8k_jacobi200}; do
  ROUND=r04 WL=$wl bash scripts/profile_round.sh || { echo "profile of $wl failed"; tail -5 gpurun_out/prof_r04_$wl/*.err gpurun_out/prof_r04_$wl/*.log | tail -30; exit 1; }
  python3 scripts/make_counters_json.py r04 $wl || exit 1
  find gpurun_out/prof_r04_$wl -name '*.csv' -size +3M -delete      # (the summaries are made; gpurun brings back at most 64 MiB)
done
mkdir -p gpurun_out/profiles_r04 && cp profiles/r04_* profiles/counters_latest.json gpurun_out/profiles_r04/
