#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['ms_per_step'], d['value'])
"; }
for B in 16 8; do for rep in 1 2; do
  for m in 0111 0112 0110 0121 0011 0101 0122; do for o in 0321 0231; do
    MERV_ENCODER_STREAM_MAP=$m MERV_ENCODER_ORDER=$o run "B$B rep $rep map $m order $o" $B
  done; done
done; done
