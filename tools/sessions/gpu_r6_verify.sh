#!/bin/bash
# Round 6: suite + smoke + default bench on the final tree (as the driver runs them).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/verify
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1800 python3 -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -3 $OUT/tests.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python3 -c "
import json
d = json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['e2e']['generated_tok_per_s'], d['e2e']['quick_start_sampled']['generated_tok_per_s'], d['e2e']['visual_path_ms'], d['e2e']['decode_ms_per_token'], d['cpu_baseline']['value'], d['parity']['pass'])
"
