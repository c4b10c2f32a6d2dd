#!/bin/bash
export MERV_HIP_LIB_AB=1  # tolerant binding for a previous build (merv_amd/_lib.py)
# ab_all.sh LIB...: ksweep (fixed cost per tile) + gemm_bench + bench.py per library, interleaved twice -> gpurun_out/ab_all.log
mkdir -p gpurun_out
OUT=gpurun_out/ab_all.log
: > $OUT
for rep in 1 2; do
  for lib in "$@"; do
    echo "== rep $rep lib=$lib ksweep" >> $OUT
    MERV_HIP_LIB=$PWD/$lib python3 tools/gemm_ksweep.py 7 2>&1 | grep fit >> $OUT
    MERV_HIP_LIB=$PWD/$lib python3 tools/gemm_ksweep.py 7 res 2>&1 | grep fit >> $OUT
    echo "== rep $rep lib=$lib gemm_bench" >> $OUT
    MERV_HIP_LIB=$PWD/$lib python3 tools/gemm_bench.py 16 0 2>&1 | grep TF >> $OUT
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
