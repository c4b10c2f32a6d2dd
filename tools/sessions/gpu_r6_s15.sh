#!/bin/bash
# Round 6, session 15: MXFP8 mode at 1 / 2 / 4 videos per step (opt-in mode; for the record), bf16 beside it.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s15
rm -rf $OUT; mkdir -p $OUT
cd $R
for B in 1 2 4; do for mode in "" "--mxfp8"; do
  timeout 300 python3 bench.py --batch $B $mode --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('B $B mode [$mode] ms_per_step', d['ms_per_step'], 'tokens/s', d['value'])
" | tee -a $OUT/mx_small_batch.txt
done; done
