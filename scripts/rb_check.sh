for wl in 1080p_jacobi1000 4k_jacobi1000 8k_jacobi1000; do
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-estimate --workload $wl --method rbgs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$wl rbgs', round(d['value']/1e3,1), d['ms_per_step'])"
done
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-estimate --workload 4k_rbsor_1e-4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('4k_rbsor', d['value'], d['ms_per_step'])"
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-estimate --workload 8k_multigrid_1e-4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('8k_mg', d['value'], d['ms_per_step'])"
