#!/bin/bash
# Round 6, session 6: suite on the current tree; remainder-launch fork; what the statistics-finalize launches cost at small
# batches (timing-only ablation); stream priority for the largest chain; the round's rocprofv3 profile set.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s6
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1800 python3 -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -5 $OUT/tests.log
timeout 900 python3 tools/probes/remainder_fork.py > $OUT/remainder_fork.txt 2> $OUT/remainder_fork.err; tail -1 $OUT/remainder_fork.txt > $OUT/remainder_fork.json; head -2 $OUT/remainder_fork.txt
for rep in 1 2; do for abl in 0 1; do for B in 1 2 4 16; do
  if [ $abl = 1 ]; then export MERV_TUNING_HOOKS=1 MERV_ABL_DOUBLE_FINALIZE=1; else export MERV_TUNING_HOOKS=1; unset MERV_ABL_DOUBLE_FINALIZE; fi
  timeout 300 python3 bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep double_finalize $abl B $B ms_per_step', d['ms_per_step'])
" | tee -a $OUT/finalize_ablation.txt
done; done; done
unset MERV_TUNING_HOOKS MERV_ABL_DOUBLE_FINALIZE
timeout 600 python3 tools/probes/stream_priority_probe.py > $OUT/stream_priority.txt 2> $OUT/stream_priority.err; tail -1 $OUT/stream_priority.txt > $OUT/stream_priority.json; head -3 $OUT/stream_priority.txt
bash tools/gpu_profile_round.sh > $OUT/profile_round.log 2>&1; tail -5 $OUT/profile_round.log
