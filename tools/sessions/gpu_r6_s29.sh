#!/bin/bash
# Round 6, session 29: sample_advance with its logits requested up front: tests, per-kernel time, the sampled e2e leg.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s29
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_decode_gpu.py -q -x -k "sample" > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
bash tools/sessions/gpu_r6_decode_trace.sh 2>&1 | grep -E "sample_advance|greedy_advance|generated_tok" | cut -c1-220
