# usage: WL=1080p_jacobi1000 bash scripts/tile_sweep.sh  -> table of tile x depth
WL=${WL:-1080p_jacobi1000}
for tile in 1 2 3 4 5 6 7 8; do for T in 4 8 12 16; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $WL --sweep-kernel 2 --tile $tile --temporal-depth $T 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('tile $tile T $T', round(d['value']/1e3,1),'Gpx-it/s', 'launch_us', round(d['roofline']['launch_us'],2), 'frac', round(d['roofline']['frac'],3))"; done; done
