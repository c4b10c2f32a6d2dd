#!/bin/bash
# Round 6, session 20: stream maps at one and two videos re-swept with the latency-critical hint in place (hooks build for MERV_ENCODER_STREAM_MAP).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s20
rm -rf $OUT; mkdir -p $OUT
cd $R
export MERV_TUNING_HOOKS=1
for rep in 1 2; do for B in 1 2; do for map in default 0123 0121 0112 0122 0111; do
  if [ $map = default ]; then unset MERV_ENCODER_STREAM_MAP; else export MERV_ENCODER_STREAM_MAP=$map; fi
  timeout 300 python3 bench.py --batch $B --steps 40 --warmup 10 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep B $B map $map ms_per_step', d['ms_per_step'])
" | tee -a $OUT/maps.txt
done; done; done
