#!/bin/bash
# Streamed attention (ViViT 3137, SigLIP 196): full key tiles as a compile-time form (no tail / half-tile branches in their body) against
# the run-time form for every tile (ab/libmerv_hip_b40.so = the library before the change): tests, attn_bench, whole step; interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/attnft; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_kernels_gpu.py tests/test_encoder_gpu.py tests/test_fulldepth_parity_gpu.py tests/test_goldens_gpu.py tests/test_backbone_variants_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest.log
LIBS="${LIBS:-merv_amd/lib/libmerv_hip.so ab/libmerv_hip_b40.so}"
bash tools/probes/ab_attn.sh $LIBS 2>&1 | tee $O/attn_bench.log
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$1: ms_per_step', d['ms_per_step'], 'gemm_ms', r['gemm_ms_per_step'], ' | '.join('%s %.2f' % (k['name'][:24], k['ms_per_step']) for k in r['by_kernel'][3:6]))
"; }
for rep in 1 2 3; do for lib in $LIBS; do
  MERV_HIP_LIB=$PWD/$lib timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | line "rep $rep $lib"
done; done | tee $O/bench.log
