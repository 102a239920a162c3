# rebuild librtdd.so on the GPU box with different register budgets for the blocked kernel and time each
cd realtimedepthdiffusion_amd/csrc
for mw in 4 3 2; do
  rm -f sweep_blocked.o
  make -s CXXFLAGS_EXTRA="-DRTDD_BLOCKED_MINWAVES=$mw" >/dev/null 2>&1
  cd ../..
  echo "== min waves/SIMD $mw"
  for cfg in "2 1 8" "2 2 8" "2 1 4"; do set -- $cfg; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --sweep-kernel $1 --tile $2 --temporal-depth $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$cfg', round(d['value']/1e3,1),'Gpx-it/s', 'launch_us', round(d['roofline']['launch_us'],2))"; done
  cd realtimedepthdiffusion_amd/csrc
done
