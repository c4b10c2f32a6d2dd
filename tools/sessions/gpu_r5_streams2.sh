#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for m in 0123 0111 0110 0112 0100; do
  MERV_ENCODER_STREAM_MAP=$m python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('B16 map $m rep $rep ms_per_step', d['ms_per_step'], d['value'])
"
done; done
for B in 1 2 4 8; do for m in 0123 0111 0110 0112; do
  MERV_ENCODER_STREAM_MAP=$m python3 bench.py --batch $B --steps 40 --warmup 10 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('B$B map $m ms_per_step', d['ms_per_step'], d['value'])
"
done; done
