for WL in 1080x1920x1000 2160x3840x200 4320x7680x50 540x960x125 270x480x250 135x240x500 67x120x1000 853x1280x125 1152x2048x100 1200x1600x100 690x960x200 624x672x250 1440x2560x60 426x640x250 910x910x200 700x560x250 256x256x250 128x128x500; do
  RTDD_DEBUG_CONFIG=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-estimate --workload $WL 2>/tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$WL', round(d['value']/1e3,1), 'Gpx-it/s', round(d['ms_per_step'],3), 'ms')"; grep -m1 "rtdd\]" /tmp/err.txt
done
