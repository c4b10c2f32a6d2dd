#!/bin/bash
# Round-5 GPU session 5: MLP chunk locality with L2-allocating output stores (does fc1's output stay in the Infinity Cache for fc2?)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s5
rm -rf $OUT; mkdir -p $OUT
cd $R
for rep in 1 2; do for lib in merv_amd/lib/libmerv_hip.so ab/libmerv_hip_plainstore.so; do
  echo "== rep $rep $lib" | tee -a $OUT/mlp_chunk.txt
  MERV_HIP_LIB=$R/$lib timeout 300 python3 tools/probes/mlp_chunk_locality.py 2>/dev/null | tee -a $OUT/mlp_chunk.txt
done; done
