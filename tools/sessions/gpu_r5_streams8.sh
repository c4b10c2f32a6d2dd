#!/bin/bash
# Stream map / enqueue order re-checked on the round's final kernels (16 videos per step): product (0111, order 0321) against neighbours
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/streams8; mkdir -p $O
run() { env "$@" timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', d['ms_per_step'])
"; }
for rep in 1 2; do
  run MERV_X=product
  run MERV_ENCODER_STREAM_MAP=0123
  run MERV_ENCODER_STREAM_MAP=0112
  run MERV_ENCODER_STREAM_MAP=0011
  run MERV_ENCODER_STREAM_MAP=0101
  run MERV_ENCODER_STREAM_MAP=0000
  run MERV_ENCODER_STREAM_MAP=0111 MERV_ENCODER_ORDER=0123
  run MERV_ENCODER_STREAM_MAP=0111 MERV_ENCODER_ORDER=0231
done | tee $O/streams.log
