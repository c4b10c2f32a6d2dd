#!/bin/bash
export MERV_HIP_LIB_AB=1  # tolerant binding for a previous build (merv_amd/_lib.py)
# ab_bench.sh LIB...: bench.py (concurrent step + one-stream GEMM leg) per library, interleaved twice; parity / CPU / e2e legs off
for rep in 1 2; do for lib in "$@"; do
  MERV_HIP_LIB=$PWD/$lib python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('rep $rep $lib: ms_per_step', d['ms_per_step'], 'gemm_ms', r['gemm_ms_per_step'], 'small-tile class', [k['ms_per_step'] for k in r['by_kernel'] if 'small' in k['name']])
"
done; done
