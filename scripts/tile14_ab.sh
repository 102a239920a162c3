for wl in 120x67_jacobi1000 240x135_jacobi500 480x270_jacobi250 960x540_jacobi125; do
  for cfg in "0 0" "9 28" "9 24" "14 28" "14 24" "14 16"; do set -- $cfg
    v=$(python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-estimate --workload $wl --tile $1 --temporal-depth $2 --persistent 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f Gpx-it/s %.4f ms' % (d['value']/1e3, d['ms_per_step']))")
    echo "$wl tile $1 depth $2: $v"
  done
done
