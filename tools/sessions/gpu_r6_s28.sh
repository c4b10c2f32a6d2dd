#!/bin/bash
# Round 6, session 28: one video -- the chains now end at 8.5 (LanguageBind), 9.3 (DINOv2), 8.8 (SigLIP): DINOv2 without the width cap / with the fast narrow form?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s28
rm -rf $OUT; mkdir -p $OUT
cd $R
export MERV_TUNING_HOOKS=1
for rep in 1 2 3; do for cfg in base nocap_dino cap128 cap128_nocap_dino cap96_nocap_dino; do
  unset MERV_BESIDE_NOCAP_M MERV_BESIDE_MAX_TILES
  case $cfg in
    nocap_dino) export MERV_BESIDE_NOCAP_M=4176;;
    cap128) export MERV_BESIDE_MAX_TILES=128;;
    cap128_nocap_dino) export MERV_BESIDE_MAX_TILES=128 MERV_BESIDE_NOCAP_M=4176;;
    cap96_nocap_dino) export MERV_BESIDE_MAX_TILES=96 MERV_BESIDE_NOCAP_M=4176;;
  esac
  timeout 300 python3 bench.py --batch 1 --steps 40 --warmup 10 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep $cfg ms_per_step', d['ms_per_step'])
" | tee -a $OUT/dino.txt
done; done
