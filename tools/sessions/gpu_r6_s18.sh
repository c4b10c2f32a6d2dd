#!/bin/bash
# Round 6, session 18: the latency-critical hint in the product (largest encoder: small tiles for sub-round launches; the others: eight-phase from 32 tiles):
# 1 / 2 / 4 / 16 videos against the round-5 library, three alternating passes; single-call latency; encoder + placement tests.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s18
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_encoder_gpu.py tests/test_fulldepth_parity_gpu.py tests/test_placement_emulated_gpu.py tests/test_vidlm_gpu.py -q -x -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
export MERV_HIP_LIB_AB=1
for rep in 1 2 3; do for lib in ab/libmerv_hip_r5.so merv_amd/lib/libmerv_hip.so; do for B in 1 2 4 16; do
  MERV_HIP_LIB=$R/$lib timeout 300 python3 bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep $lib B $B ms_per_step', d['ms_per_step'], 'tokens/s', d['value'], 'frac', d['config']['path_frac_of_mfma_peak'])
" | tee -a $OUT/sweep.txt
done; done; done
unset MERV_HIP_LIB_AB
timeout 300 python3 tools/probes/batch1_latency.py 2>/dev/null | tail -1 | tee $OUT/batch1_latency.json
