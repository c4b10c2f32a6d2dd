#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s6
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python3 -m pytest tests/test_vidlm_gpu.py tests/test_backbone_variants_gpu.py tests/test_load_vid_gpu.py -q > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -6 $OUT/tests.log
for B in 16 24 32 48 64 16; do
  timeout 400 python3 bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-prof > $OUT/sweep_b$B.json 2> $OUT/sweep_b$B.err
  python3 -c "
import json
d=json.loads(open('$OUT/sweep_b$B.json').read().strip().splitlines()[-1]); print($B, d['value'], d['ms_per_step'])"
done
