#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/attn8
MERV_ATTN_CFG=8 timeout 600 python3 -m pytest tests/test_kernels_gpu.py -q -k "atten" > gpurun_out/attn8/tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/attn8/tests.log
for rep in 1 2 3; do
  echo "== rep $rep default"; python3 tools/attn_bench.py 16 2>&1 | grep "^attn" | head -2
  echo "== rep $rep 8x1";   MERV_ATTN_CFG=8 python3 tools/attn_bench.py 16 2>&1 | grep "^attn" | head -2
done
