#!/bin/bash
# order of the encoders that share a stream (map 0111 at 16 videos: the largest alone, ranks 1-3 back to back), and two more maps
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['ms_per_step'], d['value'])
"; }
for rep in 1 2; do
  for o in 0123 0321 0231 0213 0132 0312 1230 3210; do MERV_ENCODER_ORDER=$o run "rep $rep map default order $o"; done
  MERV_ENCODER_STREAM_MAP=0100 MERV_ENCODER_ORDER=0123 run "rep $rep map 0100 (largest + two smallest | second) order 0123"
  MERV_ENCODER_STREAM_MAP=0010 MERV_ENCODER_ORDER=0123 run "rep $rep map 0010 order 0123"
done
