#!/bin/bash
# Round-5 GPU session 4: what a fully hidden epilogue could be worth -- K-loop-only (no epilogue) builds of the 256x128 ring kernel (the half-tile
# shape: 64x64 per wave, 64 accumulator registers) and of the eight-phase kernel against the product, exact-round shapes, K swept.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s4
rm -rf $OUT; mkdir -p $OUT
cd $R
for rep in 1 2; do
for lib in merv_amd/lib/libmerv_hip.so ab/libmerv_hip_noepi.so; do
  echo "== rep $rep $lib" | tee -a $OUT/ksweep.txt
  MERV_HIP_LIB=$R/$lib python3 tools/gemm_ksweep.py 4,7 2>&1 | grep "^v" | tee -a $OUT/ksweep.txt
  MERV_HIP_LIB=$R/$lib python3 tools/gemm_ksweep.py 4,7 res 2>&1 | grep "^v" | tee -a $OUT/ksweep.txt
done; done
