# the streaming sweep kernel (RTDD_OPT_SWEEP_KERNEL = 3) against the blocked one, by rows per chunk (RTDD_STREAM_ROWS)
for wl in 8k_jacobi200 4k_jacobi1000; do
  for cfg in "0 0" "3 0" "3 135" "3 100" "3 50" "3 34"; do
    set -- $cfg
    RTDD_STREAM_ROWS=$2 python bench.py --workload $wl --steps 5 --warmup 2 --no-estimate --no-cpu-baseline --sweep-kernel $1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', 'sweep kernel $1 rows per chunk $2', 'Gpx-it/s %.1f' % (d['value'] / 1e3), 'ms/step %.4f' % d['ms_per_step'], d['config'].get('kernel'))"
  done
done
