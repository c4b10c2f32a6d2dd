#!/bin/bash
# default (map by batch, smallest-first order) against the round-4 orchestration (one stream per encoder, index order), alternating, per batch; then tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --batch $2 --steps $3 --warmup 5 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['ms_per_step'], d['value'])
"; }
for B in 16 8 4 2 1; do for rep in 1 2; do
  MERV_ENCODER_STREAM_MAP=0123 MERV_ENCODER_ORDER=0123 run "B$B rep $rep round-4 orchestration" $B 20
  run "B$B rep $rep default" $B 20
  MERV_ENCODER_ORDER=0123 run "B$B rep $rep default map, index order" $B 20
done; done
timeout 1200 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -3
