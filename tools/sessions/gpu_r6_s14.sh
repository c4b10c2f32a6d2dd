#!/bin/bash
# Round 6, session 14: MXFP8 epilogues with the block maximum by DPP instead of two ds_bpermute: tests, bench --mxfp8 against the library before it.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s14
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_mxfp8_gpu.py tests/test_kernels_gpu.py tests/test_encoder_gpu.py -q -x -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for rep in 1 2; do for lib in ab/libmerv_hip_before_dpp.so merv_amd/lib/libmerv_hip.so; do
  MERV_HIP_LIB=$R/$lib timeout 400 python3 bench.py --mxfp8 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('rep $rep $lib: tokens/s', d['value'], 'ms', d['ms_per_step'], 'frac', r['frac'], 'gemm_ms', r['gemm_ms_per_step'], {k['name'][:22]: k['ms_per_step'] for k in r['by_kernel'][:3]})
" | tee -a $OUT/mx_ab.txt
done; done
