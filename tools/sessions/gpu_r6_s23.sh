#!/bin/bash
# Round 6, session 23: threaded enqueue on by default at one video: tests that run the visual path / generate(), single-call latency, default bench.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s23
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -q -x -m gpu > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -4 $OUT/tests.log
timeout 300 python3 tools/probes/batch1_latency.py 2>/dev/null | tail -1 | tee $OUT/batch1_latency.json
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python3 -c "
import json
d = json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['e2e']['generated_tok_per_s'], d['e2e']['quick_start_sampled']['generated_tok_per_s'], d['e2e']['visual_path_ms'], d['e2e']['decode_ms_per_token'], d['parity']['pass'])
"
