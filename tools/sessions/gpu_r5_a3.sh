#!/bin/bash
# Three-deep A ring of the eight-phase GEMM (A3) and the issue order of a phase's 16 MFMAs: correctness under MERV_GEMM_A3=1, per-shape
# interleaved A/B in one process, whole step per rule (MERV_GEMM_A3 = 0 never / 2 fc2-shaped / 3 N <= 1024 / 1 always), order libraries
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/a3
O=gpurun_out/a3
MERV_GEMM_A3=1 timeout 900 python3 -m pytest tests/test_gemm_variants_gpu.py tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_fulldepth_parity_gpu.py -m gpu -x -q > $O/pytest_a3.log 2>&1
echo "pytest (MERV_GEMM_A3=1) rc $?"; tail -3 $O/pytest_a3.log
timeout 600 python3 tools/probes/gemm_a3_ab.py 16 5 2>&1 | tee $O/gemm_a3_ab.log
cp gpurun_out/gemm_a3_ab.json $O/ 2>/dev/null
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$1: ms_per_step', d['ms_per_step'], 'gemm frac', r['frac'], 'gemm_ms', r['gemm_ms_per_step'], ' | '.join('%s %.2f' % (k['name'][:24], k['ms_per_step']) for k in r['by_kernel'][:3]))
"; }
for rep in 1 2; do for v in 0 2 3 1; do
  MERV_GEMM_A3=$v timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | line "rep $rep MERV_GEMM_A3=$v"
done; done | tee $O/bench_a3.log
for rep in 1 2; do for lib in merv_amd/lib/libmerv_hip.so ab/libmerv_hip_qo1.so ab/libmerv_hip_qo2.so ab/libmerv_hip_qo3.so; do
  MERV_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | line "rep $rep $lib"
done; done | tee $O/bench_qo.log
for rep in 1 2; do for lib in merv_amd/lib/libmerv_hip.so ab/libmerv_hip_qo1.so ab/libmerv_hip_qo2.so ab/libmerv_hip_qo3.so; do
  echo "== rep $rep $lib"; MERV_HIP_LIB=$PWD/$lib timeout 300 python3 tools/gemm_bench.py 16 0 2>/dev/null | grep -v embed
done; done | tee $O/gemm_bench_qo.log
