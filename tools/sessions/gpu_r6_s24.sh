#!/bin/bash
# Round 6, session 24: is the threaded enqueue's gain stable? batch1_latency.py on the PRODUCT library and on the hooks build, threads on / off.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s24
rm -rf $OUT; mkdir -p $OUT
cd $R
for rep in 1 2; do for hooks in 0 1; do for thr in 1 0; do
  if [ $hooks = 1 ]; then export MERV_TUNING_HOOKS=1; else unset MERV_TUNING_HOOKS; fi
  timeout 300 python3 tools/probes/batch1_latency.py --threads $thr 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('rep $rep hooks-build $hooks threads $thr single [median, min]', [round(x, 3) for x in d['eager_single_ms_median_min']], 'pipelined', round(d['eager_pipelined_ms'], 3), 'host', round(d['eager_host_launch_ms'], 2))
" | tee -a $OUT/lat.txt
done; done; done
