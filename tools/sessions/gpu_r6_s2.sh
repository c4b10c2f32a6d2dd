#!/bin/bash
# Round 6, session 2: bounding probe for frame-range unit chains at small batch, with 4 and 8 hardware queues.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s2
rm -rf $OUT; mkdir -p $OUT
cd $R
for q in 8 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 900 python3 tools/probes/unit_chains_probe.py 1 2 4 2>$OUT/probe_q$q.err | tee -a $OUT/unit_chains_q$q.txt | grep "^B="
done
