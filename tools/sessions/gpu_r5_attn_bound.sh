#!/bin/bash
# Resident attention (257 / 261 tokens): memory side alone (no key-tile loop), compute side alone (no K / V DMA), no output stores -- how much of
# the launch would perfect overlap of its HBM stream (Q, K, V read once, O written once: 540 MB per layer) with its compute remove?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/attnb; mkdir -p $O
for rep in 1 2; do for lib in merv_amd/lib/libmerv_hip.so ab/libmerv_hip_attnabl1.so ab/libmerv_hip_attnabl2.so ab/libmerv_hip_attnabl3.so; do
  echo "== rep $rep $lib"; ATTN_BENCH_NOCHECK=1 MERV_HIP_LIB=$PWD/$lib timeout 300 python3 tools/attn_bench.py 16 2>&1 | grep "^attn"
done; done | tee $O/attn_bound.log
