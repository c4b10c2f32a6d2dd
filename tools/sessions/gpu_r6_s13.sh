#!/bin/bash
# Round 6, session 13: what makes two-video steps 0.7 % slower than the round-5 library: the M0 clobber (hazard nops) or the run-time epilogue's
# extra store path? Variant libraries under ab/, alternating, 2 and 16 videos.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s13
rm -rf $OUT; mkdir -p $OUT
cd $R
export MERV_HIP_LIB_AB=1
for rep in 1 2 3; do for lib in ab/libmerv_hip_r5.so merv_amd/lib/libmerv_hip.so ab/libmerv_hip_nom0.so ab/libmerv_hip_r5epi.so ab/libmerv_hip_r5epi_nom0.so; do for B in 2 16; do
  MERV_HIP_LIB=$R/$lib timeout 300 python3 bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep $lib B $B ms_per_step', d['ms_per_step'])
" | tee -a $OUT/variants.txt
done; done; done
