#!/bin/bash
# A/B of library variants on the red-black sweep: scripts/ab_rbgs.sh "default rbnw"
for v in $1; do
  if [ $v = default ]; then unset RTDD_LIBRARY; else export RTDD_LIBRARY=$PWD/realtimedepthdiffusion_amd/librtdd_$v.so; fi
  for wl in 1080p_jacobi1000 4k_jacobi1000; do
    python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-estimate --workload $wl --method rbgs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v $wl rbgs', round(d['value']/1e3,1), 'Gpx-sweeps/s')"
  done
done
