#!/bin/bash
# Remaining rows of a GEMM (what complete rounds of the eight-phase kernel leave) as a partial round of the eight-phase kernel instead of small
# tiles, from MERV_REST_8PHASE_MIN_TILES tiles up: whole step + one-stream GEMM leg per threshold, two passes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/rest8; mkdir -p $O
line() { python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$1: ms_per_step', d['ms_per_step'], 'gemm frac', r['frac'], 'gemm_ms', r['gemm_ms_per_step'], ' | '.join('%s %.2f' % (k['name'][:24], k['ms_per_step']) for k in r['by_kernel'][:3]))
"; }
MERV_REST_8PHASE_MIN_TILES=16 timeout 600 python3 -m pytest tests/test_gemm_variants_gpu.py tests/test_fullsize_gpu.py tests/test_encoder_gpu.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for t in 0 4 16 48 100; do
  MERV_REST_8PHASE_MIN_TILES=$t timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | line "rep $rep min_tiles=$t"
done; done | tee $O/bench.log
