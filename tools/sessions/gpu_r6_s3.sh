#!/bin/bash
# Round 6, session 3: full GPU suite on the hooks / no-hooks builds, bench.py starting its own rank (forced distributed path on one GPU),
# the decode step's weight-prefetch bound (cold vs touched from a side stream), sub-round tile threshold re-swept at 1 / 2 / 4 videos.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s3
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -4 $OUT/tests.log
MERV_BENCH_FORCE_DISTRIBUTED=1 timeout 900 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-e2e > $OUT/forcedist_selfspawn.json 2> $OUT/forcedist_selfspawn.err; echo "forcedist self-spawn rc $?"; grep "\[bench\]" $OUT/forcedist_selfspawn.err | head -4; tail -c 600 $OUT/forcedist_selfspawn.json
timeout 600 python3 tools/probes/mall_probe.py > $OUT/decode_mall_bound.json 2> $OUT/mall.err; tail -45 $OUT/decode_mall_bound.json
export MERV_TUNING_HOOKS=1
for rep in 1 2; do for thr in 32 40 72 96; do for B in 1 2 4; do
  MERV_SUBROUND_MIN_TILES=$thr timeout 300 python3 bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep min_tiles $thr B $B ms_per_step', d['ms_per_step'])
" | tee -a $OUT/subround_sweep.txt
done; done; done
