#!/bin/bash
# the numbers DESIGN.md / README.md quote, in one GPU call: scripts/round_numbers.sh > gpurun_out/round_numbers.txt
q() { python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-estimate "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$*', '->', round(d['value']/1e3,1), d['unit'].replace('M','G',1), '%.3f ms/step' % d['ms_per_step'])"; }
for wl in 1080p_jacobi1000 4k_jacobi1000 8k_jacobi200; do q --workload $wl; q --workload $wl --method rbgs; done
for wl in 120x67_jacobi1000 240x135_jacobi500 480x270_jacobi250 960x540_jacobi125; do q --workload $wl; done
for wl in 4k_rbsor_1e-4 1080p_rbsor_1e-4 8k_multigrid_1e-4 1080p_multigrid_1e-4; do q --workload $wl; done
