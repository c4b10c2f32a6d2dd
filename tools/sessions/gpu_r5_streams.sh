#!/bin/bash
# which encoders share a side stream: fewer concurrent kernel chains = less L2 contention, more = better tail filling
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for m in 0123 0111 0011 0101 0122 0112 0120; do
  MERV_ENCODER_STREAM_MAP=$m python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('map $m rep $rep ms_per_step', d['ms_per_step'], d['value'])
"
done; done
