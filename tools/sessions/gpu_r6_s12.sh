#!/bin/bash
# Round 6, session 12: sub-round threshold 72 against the round-5 library at 1 / 2 / 4 videos (three alternating passes), then the round's profile set.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s12
rm -rf $OUT; mkdir -p $OUT
cd $R
export MERV_HIP_LIB_AB=1
for rep in 1 2 3; do for lib in ab/libmerv_hip_r5.so merv_amd/lib/libmerv_hip.so; do for B in 1 2 4; do
  MERV_HIP_LIB=$R/$lib timeout 300 python3 bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep $lib B $B ms_per_step', d['ms_per_step'], 'tokens/s', d['value'], 'frac', d['config']['path_frac_of_mfma_peak'])
" | tee -a $OUT/batch_sweep_small.txt
done; done; done
unset MERV_HIP_LIB_AB
bash tools/gpu_profile_round.sh > $OUT/profile_round.log 2>&1; tail -2 $OUT/profile_round.log | cut -c1-200
python3 -c "
import json
d = json.loads(open('$R/gpurun_out/round/bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['e2e']['generated_tok_per_s'], d['e2e']['quick_start_sampled']['generated_tok_per_s'], d['e2e']['visual_path_ms'], d['e2e']['decode_ms_per_token'])
"
