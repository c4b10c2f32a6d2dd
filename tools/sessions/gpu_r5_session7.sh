#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s7
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python3 -m pytest tests/test_vidlm_gpu.py -q > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -4 $OUT/tests.log
MERV_HIP_LIB=$R/ab/libmerv_hip_parts4.so timeout 900 python3 -m pytest tests/test_gemm_variants_gpu.py tests/test_kernels_gpu.py -q -k "gemm or epilogue" > $OUT/tests_parts4.log 2>&1; echo "pytest parts4 rc $?" >> $OUT/tests_parts4.log; tail -4 $OUT/tests_parts4.log
for rep in 1 2; do for lib in merv_amd/lib/libmerv_hip.so ab/libmerv_hip_parts4.so; do
  echo "== rep $rep $lib" | tee -a $OUT/ksweep.txt
  MERV_HIP_LIB=$R/$lib python3 tools/gemm_ksweep.py 7 res 2>&1 | grep "^v" | tee -a $OUT/ksweep.txt
  MERV_HIP_LIB=$R/$lib python3 tools/gemm_ksweep.py 7 2>&1 | grep "^v" | tee -a $OUT/ksweep.txt
  MERV_HIP_LIB=$R/$lib timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'])" | tee -a $OUT/ksweep.txt
done; done
