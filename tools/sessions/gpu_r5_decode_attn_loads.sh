#!/bin/bash
# Decode attention (split launch): everything that hangs on *pos requested at once, next trip of cache rows in flight under the multiply
# of this one -- against the library before (LIBS), tests + per-class times + e2e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/dattn; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_decode_gpu.py tests/test_generate_gpu.py tests/test_vidlm_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -2 $O/pytest.log
LIBS="${LIBS:-merv_amd/lib/libmerv_hip.so ab/libmerv_hip_head5.so}"
for rep in 1 2; do for lib in $LIBS; do
  echo "== rep $rep $lib"; MERV_HIP_LIB=$PWD/$lib timeout 300 python3 tools/probes/decode_kernels.py 2>/dev/null | tail -1
done; done | tee $O/decode_kernels.log
for rep in 1 2; do for lib in $LIBS; do
  MERV_HIP_LIB=$PWD/$lib timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); e = d.get('e2e') or {}
        print('rep $rep $lib: ms_per_step', d['ms_per_step'], 'e2e', e.get('generated_tok_per_s'), 'decode ms', e.get('decode_ms_per_token'))
"
done; done | tee $O/bench_e2e.log
