#!/bin/bash
# GPU box: kernel trace of three training steps (reduced LLM depth keeps the trace small; the visual-path kernels and
# their backward are the full-size ones). Output: gpurun_out/train/kernel_stats.csv
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/train
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/train_bench.py --llm-layers 2 --steps 3 --warmup 1 > $OUT/train_bench.json 2> $OUT/trace.err
tail -1 $OUT/train_bench.json
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
grep -E "merv::" $OUT/kernel_stats.csv | cut -c1-200
rm -rf $OUT/trace
