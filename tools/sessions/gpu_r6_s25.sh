#!/bin/bash
# Round 6, session 25: stream priority for the largest chain re-measured with the stream bound to its queue at creation.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s25
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python3 tools/probes/stream_priority_probe.py > $OUT/stream_priority.txt 2> $OUT/err.log; tail -1 $OUT/stream_priority.txt > $OUT/stream_priority.json; head -3 $OUT/stream_priority.txt
