#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['ms_per_step'], d['value'])
"; }
for rep in 1 2; do
  run "B16 rep $rep default" 16
  MERV_ENCODER_STREAM_PRIO=01 run "B16 rep $rep shared chain high priority" 16
  MERV_ENCODER_STREAM_PRIO=10 run "B16 rep $rep largest encoder high priority" 16
done
