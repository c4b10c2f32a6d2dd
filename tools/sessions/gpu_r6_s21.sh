#!/bin/bash
# Round 6, session 21: one / two videos -- the chains beside the critical one capped in WIDTH as well (launches of more than MERV_BESIDE_MAX_TILES tiles split
# into consecutive launches). Hooks build, alternating; bits checked by the encoder batch-invariance test under the cap.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s21
rm -rf $OUT; mkdir -p $OUT
cd $R
export MERV_TUNING_HOOKS=1
for rep in 1 2 3; do for B in 1 2; do for cap in 0 192 160 128 96; do
  if [ $cap = 0 ]; then unset MERV_BESIDE_MAX_TILES; else export MERV_BESIDE_MAX_TILES=$cap; fi
  timeout 300 python3 bench.py --batch $B --steps 40 --warmup 10 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep B $B cap $cap ms_per_step', d['ms_per_step'])
" | tee -a $OUT/cap.txt
done; done; done
