#!/bin/bash
export MERV_HIP_LIB_AB=1  # tolerant binding for a previous build (merv_amd/_lib.py)
# ab_ksweep.sh LIB...: tools/gemm_ksweep.py (fixed cost per 256x256 tile round, with and without a residual) for each library, twice, interleaved
mkdir -p gpurun_out
OUT=gpurun_out/ab_ksweep.log
: > $OUT
for rep in 1 2; do
  for lib in "$@"; do
    echo "== rep $rep lib=$lib" >> $OUT
    MERV_HIP_LIB=$PWD/$lib python3 tools/gemm_ksweep.py 7 >> $OUT 2>&1
    MERV_HIP_LIB=$PWD/$lib python3 tools/gemm_ksweep.py 7 res >> $OUT 2>&1
  done
done
cat $OUT
