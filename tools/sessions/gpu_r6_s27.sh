#!/bin/bash
# Round 6, session 27: when each chain ends inside a one- / two-video step (which chain is critical under the round's launch policy).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s27
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 300 python3 tools/probes/chain_end_times.py 1 2 4 > $OUT/chain_ends.txt 2> $OUT/err.log; tail -1 $OUT/chain_ends.txt > $OUT/chain_ends.json; head -3 $OUT/chain_ends.txt; tail -2 $OUT/err.log
