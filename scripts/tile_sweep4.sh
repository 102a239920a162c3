# round 2, new sweep body: is the automatic (tile, depth) still the best?  4K (launch per block) and 1080p (persistent)
for WL in 4k_jacobi1000 1080p_jacobi1000; do
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-estimate --workload $WL 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$WL auto', round(d['value']/1e3,1),'Gpx-it/s', d['config']['tile'], d['config']['temporal_depth'], d['config']['persistent'])"
  for tile in 2 3 4 5 6 8 12 13; do for T in 6 8 12 16; do python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-estimate --workload $WL --sweep-kernel 2 --tile $tile --temporal-depth $T 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$WL tile $tile T $T', round(d['value']/1e3,1),'Gpx-it/s persistent', d['config']['persistent'])"; done; done
done
