#!/bin/bash
# Round 6, session 17: one video -- the small-tile form for the sub-round launches of SOME chains only (the largest: it ends the step), round 5's
# eight-phase form for the others (fewer CUs per launch: more room for the critical chain). Hooks build, alternating.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s17
rm -rf $OUT; mkdir -p $OUT
cd $R
export MERV_TUNING_HOOKS=1
for rep in 1 2 3; do for only in "" "4112" "4112,4176" "4112,3137,3136" "4176,3137,3136" "3137,3136"; do
  if [ -n "$only" ]; then export MERV_SUBROUND_ONLY_M=$only; else unset MERV_SUBROUND_ONLY_M; fi
  timeout 300 python3 bench.py --batch 1 --steps 40 --warmup 10 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep small tiles only for M in [$only] ms_per_step', d['ms_per_step'])
" | tee -a $OUT/only_m.txt
done; done
