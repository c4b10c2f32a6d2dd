#!/bin/bash
# Round 6, session 16: the new batch-invariance tests (bf16 and MXFP8).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s16
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_encoder_gpu.py tests/test_mxfp8_gpu.py -q -m gpu -k "same_bits" > $OUT/tests.log 2>&1; tail -15 $OUT/tests.log
