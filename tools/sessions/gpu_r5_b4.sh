#!/bin/bash
# Eight-phase GEMM, which phase requests which DMA quarter: B4 (B-late with B-early in phase 4, nothing in phase 1) against the
# two-pieces-per-phase placement (ab/libmerv_hip_b40.so = the same source with -DMERV_GEMM_B4=0): tests on the new form, then A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/b4; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gemm_variants_gpu.py tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_fulldepth_parity_gpu.py tests/test_goldens_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.log
LIBS="${LIBS:-merv_amd/lib/libmerv_hip.so ab/libmerv_hip_b40.so}"
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
  echo "== rep $rep $lib"; MERV_HIP_LIB=$PWD/$lib timeout 300 python3 tools/gemm_ksweep.py 7 2>/dev/null | tail -2
done; done | tee $O/ksweep.log
