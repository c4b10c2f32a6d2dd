#!/bin/bash
export MERV_HIP_LIB_AB=1  # tolerant binding for a previous build (merv_amd/_lib.py)
# ab_gemm.sh LIB...: per library (interleaved, twice): tools/gemm_bench.py 16 0 (the encoder stack's shapes at 16 videos, tile choice
# of the library) and bench.py (whole step); -> gpurun_out/ab_gemm.log
mkdir -p gpurun_out
OUT=gpurun_out/ab_gemm.log
: > $OUT
for rep in 1 2; do
  for lib in "$@"; do
    echo "== rep $rep lib=$lib gemm_bench" >> $OUT
    MERV_HIP_LIB=$PWD/$lib python3 tools/gemm_bench.py 16 0 >> $OUT 2>&1
  done
done
for rep in 1 2; do
  for lib in "$@"; do
    echo "== rep $rep lib=$lib bench" >> $OUT
    MERV_HIP_LIB=$PWD/$lib python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('ms_per_step', d['ms_per_step'], 'gemm frac', r['frac'], 'gemm_ms', r['gemm_ms_per_step'], ' | '.join('%s %.2f' % (k['name'][:24], k['ms_per_step']) for k in r['by_kernel'][:3]))
" >> $OUT
  done
done
cat $OUT
