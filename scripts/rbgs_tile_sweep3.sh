for wl in 540x960x240 270x480x240 1080x1920x240 2160x3840x240; do
  for cfg in "0 0" "1 4" "1 8" "2 4" "2 8"; do set -- $cfg
    v=$(python bench.py --method rbgs --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --tile $1 --temporal-depth $2 2>/dev/null | python -c "import json,sys; print('%.0f' % (json.loads(sys.stdin.readline())['value']/1e3))")
    echo "$wl tile $1 depth $2: $v Gpx-sweeps/s"
  done
done
