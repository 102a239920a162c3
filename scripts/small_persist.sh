for wl in 67x120x1000 135x240x500; do
  for cfg in "0 0 -1" "9 24 0" "9 24 1" "9 16 1" "9 12 1" "11 24 1" "8 16 1" "4 8 1"; do set -- $cfg
    v=$(python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-estimate --workload $wl --tile $1 --temporal-depth $2 --persistent $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f Gpx-it/s %.3f ms' % (d['value']/1e3, d['ms_per_step']))")
    echo "$wl tile $1 depth $2 persistent $3: $v"
  done
done
