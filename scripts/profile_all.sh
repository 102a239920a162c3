#!/bin/bash
# A round's profiles, part 1: the bench workloads (kernel trace + FETCH / WRITE / two SQ passes each), then the summaries under profiles/.
# usage (on the GPU box): ROUND=r05 bash scripts/profile_all.sh "1080p_jacobi1000 4k_jacobi1000 8k_jacobi200"
set -o pipefail
cd $GRAFT_REPO_ROOT
R=${ROUND:-r05}
for wl in ${1:-1080p_jacobi1000 4k_jacobi1000 8k_jacobi200}; do
  ROUND=$R WL=$wl bash scripts/profile_round.sh || { echo "profile of $wl failed"; tail -5 gpurun_out/prof_${R}_$wl/*.err gpurun_out/prof_${R}_$wl/*.log | tail -30; exit 1; }
  python3 scripts/make_counters_json.py $R $wl || exit 1
  find gpurun_out/prof_${R}_$wl -name '*.csv' -size +3M -delete      # (the summaries are made; gpurun brings back at most 64 MiB)
done
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_* profiles/counters_latest.json gpurun_out/profiles_$R/
