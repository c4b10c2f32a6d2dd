#!/bin/bash
# Round 6, session 7: what the statistics-finalize launches cost at small batches (each issued twice, hooks build: the slow-down is their cost).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s7
rm -rf $OUT; mkdir -p $OUT
cd $R
export MERV_TUNING_HOOKS=1
for rep in 1 2 3; do for abl in 0 1; do for B in 1 2 4 16; do
  if [ $abl = 1 ]; then export MERV_ABL_DOUBLE_FINALIZE=1; else unset MERV_ABL_DOUBLE_FINALIZE; fi
  timeout 300 python3 bench.py --batch $B --steps 30 --warmup 12 --no-cpu-baseline --no-e2e --no-prof 2>$OUT/err.log | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep double_finalize $abl B $B ms_per_step', d['ms_per_step'])
" | tee -a $OUT/finalize_ablation.txt
done; done; done
tail -3 $OUT/err.log
