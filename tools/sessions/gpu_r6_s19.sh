#!/bin/bash
# Round 6, session 19: CU-masked streams for the non-critical chains at 1 / 2 videos.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s19
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python3 tools/probes/cumask_chains_probe.py > $OUT/cumask_chains.txt 2> $OUT/cumask.err; tail -1 $OUT/cumask_chains.txt > $OUT/cumask_chains.json; grep -v "^{" $OUT/cumask_chains.txt | head -12; tail -2 $OUT/cumask.err
