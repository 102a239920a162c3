# red-black kernel: tile shape x sweeps per launch x image size (bench.py --method rbgs)
for wl in 1080x1920x200 2160x3840x200 4320x7680x100; do
  for tile in 1 2; do for d in 2 3 4 6 8; do
    v=$(python bench.py --method rbgs --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --tile $tile --temporal-depth $d 2>/dev/null | python -c "import json,sys; print('%.0f' % (json.loads(sys.stdin.readline())['value']/1e3))")
    echo "$wl tile $tile depth $d: $v Gpx-sweeps/s"
  done; done
done
