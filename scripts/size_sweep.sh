# explore tile/depth/persistent choices on awkward sizes
for WL in 853x1280x125 1152x2048x100 1200x1600x100 690x960x200 624x672x250 1440x2560x60 426x640x250; do
  echo "== $WL"
  for cfg in "0 0 1" "4 8 1" "4 8 0" "9 8 0" "9 8 1" "6 8 0" "8 8 0" "8 8 1" "12 8 1" "5 8 1" "5 8 0"; do set -- $cfg
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-estimate --workload $WL --tile $1 --temporal-depth $2 --persistent $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   tile $1 T $2 persist $3:', round(d['value']/1e3,1), 'Gpx-it/s', 'sweeps/launch', round(d['config']['sweeps_per_launch'],1))"
  done
done
