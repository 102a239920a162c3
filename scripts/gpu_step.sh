#!/bin/bash
# run one GPU step under its own timeout, appending progress to gpurun_out/steps.log (so a long call never looks hung)
# usage: scripts/gpu_step.sh SECONDS cmd...
T=$1; shift
echo "== $(date +%T) $*" | tee -a gpurun_out/steps.log
timeout -k 5 $T "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/steps.log
rc=${PIPESTATUS[0]}
echo "-- rc=$rc" | tee -a gpurun_out/steps.log
exit $rc
