#!/bin/bash
# Eight-phase GEMM, whole-tile form of the LDS epilogue (all rows valid and the residual known at compile time: unconditional stores, exact vmcnt
# counts, both parts' residual rows requested up front) against the general form (ab/libmerv_hip_av0.so = same source, -DMERV_GEMM_ALLVALID=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/whole; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gemm_variants_gpu.py tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_fulldepth_parity_gpu.py tests/test_goldens_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.log
LIBS="${LIBS:-merv_amd/lib/libmerv_hip.so ab/libmerv_hip_av0.so}"
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$1: ms_per_step', d['ms_per_step'], 'gemm frac', r['frac'], 'gemm_ms', r['gemm_ms_per_step'], ' | '.join('%s %.2f' % (k['name'][:24], k['ms_per_step']) for k in r['by_kernel'][:3]))
"; }
for rep in 1 2 3; do for lib in $LIBS; do
  MERV_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | line "rep $rep $lib"
done; done | tee $O/bench.log
for rep in 1 2; do for lib in $LIBS; do
  echo "== rep $rep $lib"; MERV_HIP_LIB=$PWD/$lib timeout 300 python3 tools/gemm_ksweep.py 7 res 2>/dev/null | tail -1; GEMM_BENCH_INPLACE=1 MERV_HIP_LIB=$PWD/$lib timeout 300 python3 tools/gemm_bench.py 16 0 2>/dev/null | grep "proj\|fc2"
done; done | tee $O/ksweep.log
