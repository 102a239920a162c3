#!/bin/bash
# Build an A/B variant of the library: scripts/build_variant.sh NAME "-DFOO=1 ..." [file.hip ...]
# -> realtimedepthdiffusion_amd/librtdd_NAME.so (the listed .hip files recompiled with the defines, the rest reused);
# select it at run time with RTDD_LIBRARY=realtimedepthdiffusion_amd/librtdd_NAME.so (a developer knob of the Python mirror).
set -e
NAME=$1; DEFS=$2; shift 2
FILES=${@:-sweep_blocked.hip}
cd "$(dirname "$0")/../realtimedepthdiffusion_amd/csrc"
make -j4 >/dev/null
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fvisibility=hidden -I../../include -I."
OBJS=""
for f in solver_kernels sweep_blocked rbgs_blocked multigrid image_kernels effect_kernels cascade api cascade_api dropin; do
    ext=hip; [ -f $f.cpp ] && ext=cpp
    if echo " $FILES " | grep -q " $f.$ext "; then
        /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $DEFS -c $f.$ext -o /tmp/${f}_$NAME.o
        OBJS="$OBJS /tmp/${f}_$NAME.o"
    else
        OBJS="$OBJS $f.o"
    fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librtdd_$NAME.so $OBJS
echo built librtdd_$NAME.so
