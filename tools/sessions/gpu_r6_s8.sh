#!/bin/bash
# Round 6, session 8: does an any-order launch overlap two kernels of one stream on gfx950 (probe); single-call latency of the one-video visual path
# by enqueue order of the encoders.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s8
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 120 ./ab/anyorder_probe > $OUT/anyorder_probe.json 2> $OUT/anyorder.err; cat $OUT/anyorder_probe.json
export MERV_TUNING_HOOKS=1
for rep in 1 2; do for ord in 0321 0123 0132 1023 0312; do
  MERV_ENCODER_ORDER=$ord timeout 300 python3 tools/probes/batch1_latency.py 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('rep $rep order $ord single-call median/min ms', d['eager_single_ms_median_min'], 'pipelined', d['eager_pipelined_ms'], 'host enqueue', round(d['eager_host_launch_ms'], 2))
" | tee -a $OUT/order_b1.txt
done; done
