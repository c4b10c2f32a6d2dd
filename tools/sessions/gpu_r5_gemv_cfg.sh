#!/bin/bash
# Decode GEMV: rows per wave x chunks per trip per launch class (fused-norm q/k/v, the gate-up pair, lm_head), tools/probes/decode_kernels.py
# per configuration (one process each: the hooks are read once), twice
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/gemv; mkdir -p $O
: > $O/gemv_cfg.log
for rep in 1 2; do
  for cfg in "base" "MERV_GEMV_CFG_NORM=18" "MERV_GEMV_CFG_NORM=14" "MERV_GEMV_CFG_PAIR=18" "MERV_GEMV_CFG_PAIR=14" "MERV_GEMV_CFG_BIG=18" "MERV_GEMV_CFG_NORM=18 MERV_GEMV_CFG_PAIR=18 MERV_GEMV_CFG_BIG=18"; do
    echo "== rep $rep $cfg" | tee -a $O/gemv_cfg.log
    if [ "$cfg" = "base" ]; then timeout 300 python3 tools/probes/decode_kernels.py 2>/dev/null | tail -1 | tee -a $O/gemv_cfg.log
    else env $cfg timeout 300 python3 tools/probes/decode_kernels.py 2>/dev/null | tail -1 | tee -a $O/gemv_cfg.log; fi
  done
done
