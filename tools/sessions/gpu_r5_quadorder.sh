#!/bin/bash
# Issue order of the 16 MFMAs of an eight-phase phase: product (0) against orders 2, 4, 5, 6, 7 (tools/probes/gemm_probe_hooks.h,
# MERV_ABL_QUAD_ORDER), whole step + one-stream GEMM leg, libraries interleaved, three passes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/qo
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$1: ms_per_step', d['ms_per_step'], 'gemm frac', r['frac'], 'gemm_ms', r['gemm_ms_per_step'], ' | '.join('%s %.2f' % (k['name'][:24], k['ms_per_step']) for k in r['by_kernel'][:3]))
"; }
LIBS="${LIBS:-merv_amd/lib/libmerv_hip.so ab/libmerv_hip_qo2.so ab/libmerv_hip_qo4.so ab/libmerv_hip_qo5.so ab/libmerv_hip_qo6.so ab/libmerv_hip_qo7.so}"
for rep in 1 2 3; do for lib in $LIBS; do
  MERV_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | line "rep $rep $lib"
done; done | tee gpurun_out/qo/bench_qo.log
