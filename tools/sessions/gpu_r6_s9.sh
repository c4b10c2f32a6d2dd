#!/bin/bash
# Round 6, session 9: one host thread per encoder chain at one video per call (single-call latency), graph replay re-measured.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s9
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python3 tools/probes/threaded_enqueue_probe.py > $OUT/threaded_enqueue.txt 2> $OUT/threaded.err; tail -1 $OUT/threaded_enqueue.txt > $OUT/threaded_enqueue.json; head -2 $OUT/threaded_enqueue.txt; tail -3 $OUT/threaded.err
timeout 300 python3 tools/probes/batch1_latency.py 2>/dev/null | tail -1 | tee $OUT/batch1_latency.json
