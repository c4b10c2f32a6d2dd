#!/bin/bash
# Plain decode GEMV with the block's x as an LDS image and the wave's whole row in flight (MERV_GEMV_XLDS probe hook, see launch_decode_gemv) against the product form: bits, tests under the hook, per-class times, e2e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/xlds; mkdir -p $O
HOOKS="${HOOKS:-0 1 3 4}"
for h in $HOOKS; do echo "== bits XLDS=$h"; MERV_GEMV_XLDS=$h timeout 300 python3 tools/probes/gemv_bits.py 2>&1 | tail -15; done | tee $O/bits.log
MERV_GEMV_XLDS=${TEST_HOOK:-4} timeout 1200 python3 -m pytest tests/test_decode_gpu.py tests/test_generate_gpu.py -m gpu -x -q > $O/pytest_xlds2.log 2>&1
echo "pytest (XLDS ${TEST_HOOK:-4}) rc $?"; tail -2 $O/pytest_xlds2.log
for rep in 1 2; do for h in $HOOKS; do
  echo "== rep $rep XLDS=$h"; MERV_GEMV_XLDS=$h timeout 300 python3 tools/probes/decode_kernels.py 2>/dev/null | tail -1
done; done | tee $O/decode_kernels.log
for rep in 1 2; do for h in $HOOKS; do
  MERV_GEMV_XLDS=$h timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); e = d.get('e2e') or {}
        print('rep $rep XLDS=$h: e2e', e.get('generated_tok_per_s'), 'decode ms', e.get('decode_ms_per_token'))
"
done; done | tee $O/bench_e2e.log
