for r in 0 1 0 1; do
  for wl in 4k_jacobi1000 8k_jacobi200; do
    v=$(RTDD_XCD_ALL=$r python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-estimate --workload $wl 2>/dev/null | python -c "import json,sys; print('%.1f' % (json.loads(sys.stdin.readline())['value']/1e3))")
    echo "xcd_all $r $wl: $v Gpx-it/s"
  done
done
