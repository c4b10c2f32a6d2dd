#!/bin/bash
# tile-order group size (m-tiles per column sweep) of every GEMM launch under the two-chain stream map
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py --batch 16 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['ms_per_step'], d['value'])
"; }
for rep in 1 2; do
  run "rep $rep group 4 (default)"
  MERV_GEMM_GROUP_M=2 run "rep $rep group 2"
  MERV_GEMM_GROUP_M=8 run "rep $rep group 8"
  MERV_GEMM_GROUP_M=16 run "rep $rep group 16"
done
