for wl in 1080x1920x240 2160x3840x240 4320x7680x120 540x960x240; do
  for tile in 2; do for d in 8 10 12 16 20; do
    v=$(python bench.py --method rbgs --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --tile $tile --temporal-depth $d 2>/dev/null | python -c "import json,sys; print('%.0f' % (json.loads(sys.stdin.readline())['value']/1e3))")
    echo "$wl tile $tile depth $d: $v Gpx-sweeps/s"
  done; done
done
