#!/bin/bash
# Round 6, session 26: two to eight videos -- what the complete rounds leave over taken by the eight-phase kernel as a last partial round when it is at least
# MERV_REST8_MIN_TILES tiles (instead of the small tiles). Hooks build, alternating.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s26
rm -rf $OUT; mkdir -p $OUT
cd $R
export MERV_TUNING_HOOKS=1
for rep in 1 2; do for B in 2 3 4 8 16; do for thr in 0 64 128 192; do
  if [ $thr = 0 ]; then unset MERV_REST8_MIN_TILES; else export MERV_REST8_MIN_TILES=$thr; fi
  timeout 300 python3 bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep B $B rest8 >= $thr ms_per_step', d['ms_per_step'])
" | tee -a $OUT/rest8.txt
done; done; done
