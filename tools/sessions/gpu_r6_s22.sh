#!/bin/bash
# Round 6, session 22: width cap for the chains beside the critical one, wide launches (>= 8 column tiles) only: 1 / 2 / 3 / 4 videos.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s22
rm -rf $OUT; mkdir -p $OUT
cd $R
export MERV_TUNING_HOOKS=1 MERV_BESIDE_WIDE_N=8
for rep in 1 2 3; do for B in 1 2 3 4; do for cap in 0 160 128 112; do
  if [ $cap = 0 ]; then unset MERV_BESIDE_MAX_TILES; else export MERV_BESIDE_MAX_TILES=$cap; fi
  timeout 300 python3 bench.py --batch $B --steps 40 --warmup 10 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep B $B cap $cap (wide launches only) ms_per_step', d['ms_per_step'])
" | tee -a $OUT/cap_wide.txt
done; done; done
