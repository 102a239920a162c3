# SQ counter passes (VALU issue, LDS stalls, waits) for the kernels the round's roofline record names.
# Counters only: --pmc is never combined with a trace domain other than --kernel-trace.
# usage: TAG=r02a bash scripts/profile_sq.sh     (outputs under gpurun_out/sq_$TAG, summary gpurun_out/sq_$TAG/summary.json)
export TMPDIR=/tmp
TAG=${TAG:-r02}
OUT=gpurun_out/sq_$TAG; mkdir -p $OUT
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM"
run() {   # name, python args...
    local name=$1; shift
    rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $OUT -o ${name}_p1 -- python3 "$@" > $OUT/${name}_p1.log 2>&1 || return 1
    rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d $OUT -o ${name}_p2 -- python3 "$@" > $OUT/${name}_p2.log 2>&1 || return 1
    echo "done $name"
}
B="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-estimate"
run 1080p_jacobi $B --workload 1080p_jacobi1000 &&
run 4k_jacobi $B --workload 4k_jacobi1000 &&
run 1080p_rbgs $B --workload 1080x1920x400 --method rbgs &&
run 4k_rbgs $B --workload 2160x3840x200 --method rbgs &&
run 8k_mg scripts/mg_profile.py 4320 7680 4 &&
python3 scripts/sq_summary.py $OUT > $OUT/summary.json && cat $OUT/summary.json | head -c 6000
