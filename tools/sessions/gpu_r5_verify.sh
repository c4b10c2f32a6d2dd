#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/verify
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -3 $OUT/tests.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; head -c 600 $OUT/bench.json; echo
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/verify/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic_source"][:50], d["parity"]["pass"], d["cpu_baseline"]["value"], d["config"]["e2e_gen_tok_s"])
print([(k["name"][:20], k["frac"], k.get("frac_at_clock")) for k in d["roofline"]["by_kernel"][:2]])
PY
