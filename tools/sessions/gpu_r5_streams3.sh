#!/bin/bash
# the batch-dependent encoder -> stream map (default) against one stream per encoder (MERV_ENCODER_STREAM_MAP=0123), alternating; then the path's tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/streams
for rep in 1 2 3; do for m in 0123 default; do
  if [ $m = default ]; then unset MERV_ENCODER_STREAM_MAP; else export MERV_ENCODER_STREAM_MAP=$m; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('B16 map $m rep $rep ms_per_step', d['ms_per_step'], d['value'])
"
done; done
for B in 1 2 4 8; do for m in 0123 default; do
  if [ $m = default ]; then unset MERV_ENCODER_STREAM_MAP; else export MERV_ENCODER_STREAM_MAP=$m; fi
  python3 bench.py --batch $B --steps 40 --warmup 10 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('B$B map $m ms_per_step', d['ms_per_step'], d['value'])
"
done; done
unset MERV_ENCODER_STREAM_MAP
timeout 900 python3 -m pytest tests/test_vidlm_gpu.py tests/test_fulldepth_parity_gpu.py tests/test_placement_emulated_gpu.py tests/test_fullsize_gpu.py tests/test_merv_forward_gpu.py tests/test_train_gpu.py -q 2>&1 | tail -3
