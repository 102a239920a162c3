#!/bin/bash
# the driver's bench command from fresh processes, with and without the untimed clock ramp
for f in "" "--no-clock-ramp" "" "--no-clock-ramp"; do
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-estimate $f 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags [$f] ->', 'Gpx-it/s %.1f' % (d['value']/1e3), 'ms %.3f' % d['ms_per_step'], 'launch_us %.1f' % d['roofline']['launch_us'])"
sleep 3
done
