#!/bin/bash
# Decode GEMV with the block-level normalised-x LDS image and the new default configurations: decode / generate / vidlm tests, per-class
# times, e2e leg of bench.py against the library before the change (ab/libmerv_hip_b40.so), interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/gemv; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_decode_gpu.py tests/test_generate_gpu.py tests/test_vidlm_gpu.py tests/test_load_vid_gpu.py -m gpu -x -q > $O/pytest_xn2.log 2>&1
echo "pytest rc $?"; tail -3 $O/pytest_xn2.log
for rep in 1 2; do for lib in merv_amd/lib/libmerv_hip.so ab/libmerv_hip_b40.so; do
  echo "== rep $rep $lib"; MERV_HIP_LIB=$PWD/$lib timeout 300 python3 tools/probes/decode_kernels.py 2>/dev/null | tail -1
done; done | tee $O/decode_kernels_xn2.log
for rep in 1 2; do for lib in merv_amd/lib/libmerv_hip.so ab/libmerv_hip_b40.so; do
  MERV_HIP_LIB=$PWD/$lib timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d['config']
        print('rep $rep $lib: ms_per_step', d['ms_per_step'], 'e2e', {k: v for k, v in c.items() if 'e2e' in k or 'decode' in k or 'prefill' in k})
"
done; done | tee $O/bench_e2e_xn2.log
