# after the XCD-aware placement: is the automatic (tile, depth) still the best at 4K / 8K?  (launch-per-block path)
for WL in 4k_jacobi1000 8k_jacobi200; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-estimate --workload $WL 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$WL auto', round(d['value']/1e3,1),'Gpx-it/s', d['config'])"
  for tile in 2 3 4 5 6 8 12; do for T in 4 6 8 12 16; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-estimate --workload $WL --sweep-kernel 2 --tile $tile --temporal-depth $T 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$WL tile $tile T $T', round(d['value']/1e3,1),'Gpx-it/s')"; done; done
done
