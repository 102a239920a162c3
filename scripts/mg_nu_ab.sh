for v in default nu1 nu3; do
  if [ $v = default ]; then unset RTDD_LIBRARY; else export RTDD_LIBRARY=$PWD/realtimedepthdiffusion_amd/librtdd_$v.so; fi
  for wl in 8k_multigrid_1e-4 1080p_multigrid_1e-4; do
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-estimate --workload $wl 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v $wl', '%.3f ms' % d['ms_per_step'], d['config'].get('converged'))"
  done
done
