#!/bin/bash
# timing-only ablations of k_defocus_tile (results wrong): 1 no row scan, 2 no lookups, 4 no unpack, 7 all three
for v in default dt1 dt2 dt4 dt7; do
  if [ $v = default ]; then unset RTDD_LIBRARY; else export RTDD_LIBRARY=$PWD/realtimedepthdiffusion_amd/librtdd_$v.so; fi
  echo "== $v"; python3 scripts/defocus_paths.py 2>/dev/null | head -2
done
