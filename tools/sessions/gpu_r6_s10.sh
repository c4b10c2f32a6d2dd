#!/bin/bash
# Round 6, session 10: rocprofv3 kernel trace of the MXFP8 step and of the one-video step (one stream: per-kernel durations).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s10
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_mx -- python3 $R/bench.py --mxfp8 --sequential --steps 6 --warmup 2 --no-prof --no-cpu-baseline --no-e2e > $OUT/trace_mx.json 2> $OUT/trace_mx.err
cp $(find $OUT/trace_mx -name "*kernel_stats.csv" | head -1) $OUT/mxfp8_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_b1 -- python3 $R/bench.py --batch 1 --sequential --steps 20 --warmup 5 --no-prof --no-cpu-baseline --no-e2e > $OUT/trace_b1.json 2> $OUT/trace_b1.err
cp $(find $OUT/trace_b1 -name "*kernel_stats.csv" | head -1) $OUT/batch1_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
head -30 $OUT/mxfp8_kernel_stats.csv | cut -c1-200
