#!/bin/bash
# Decode GEMV with the block-level normalised-x LDS image: decode / generate tests, then the per-class configuration sweep again
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/gemv; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_decode_gpu.py tests/test_generate_gpu.py -m gpu -x -q > $O/pytest_xn.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest_xn.log
: > $O/gemv_cfg_xn.log
for rep in 1 2; do
  for cfg in "base" "MERV_GEMV_CFG_NORM=18" "MERV_GEMV_CFG_NORM=14" "MERV_GEMV_CFG_PAIR=18" "MERV_GEMV_CFG_PAIR=14" "MERV_GEMV_CFG_NORM=18 MERV_GEMV_CFG_PAIR=14" "MERV_GEMV_CFG_NORM=18 MERV_GEMV_CFG_PAIR=18 MERV_GEMV_CFG_BIG=18"; do
    echo "== rep $rep $cfg" | tee -a $O/gemv_cfg_xn.log
    if [ "$cfg" = "base" ]; then timeout 300 python3 tools/probes/decode_kernels.py 2>/dev/null | tail -1 | tee -a $O/gemv_cfg_xn.log
    else env $cfg timeout 300 python3 tools/probes/decode_kernels.py 2>/dev/null | tail -1 | tee -a $O/gemv_cfg_xn.log; fi
  done
done
