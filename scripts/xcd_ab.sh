# persistent kernel: tile -> XCD placement A/B (RTDD_XCD_REMAP 0 = round robin, 1 = bands of consecutive tiles per XCD; a third variant, filling one XCD first, measured like 1)
for r in 0 1; do
  for wl in 1080p_jacobi1000 960x540x1000 480x270x1000 240x135x1000; do
    v=$(RTDD_XCD_REMAP=$r python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-estimate --workload $wl 2>/dev/null | python -c "import json,sys; print('%.1f' % (json.loads(sys.stdin.readline())['value']/1e3))")
    echo "xcd mode $r $wl: $v Gpx-it/s"
  done
  RTDD_XCD_REMAP=$r python scripts/estimate_bench.py 1080 1920 2>/dev/null | head -1
done
