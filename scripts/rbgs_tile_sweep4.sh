# red-black kernel after round 2's poll change: tile shape x sweeps per launch x image size (is the automatic choice still the best?)
for wl in 1080x1920x240 2160x3840x240 4320x7680x96; do
  for cfg in "0 0" "1 4" "1 6" "1 8" "1 12" "2 4" "2 6" "2 8" "2 12" "2 16"; do set -- $cfg
    v=$(python3 bench.py --method rbgs --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-estimate --tile $1 --temporal-depth $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.0f Gpx-sweeps/s persistent %s' % (d['value']/1e3, d['config'].get('persistent')))")
    echo "$wl tile $1 depth $2: $v"
  done
done
