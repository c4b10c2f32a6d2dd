#!/bin/bash
# Round 6, session 5: full suite (MX forms, device-side sampling), default bench line (both e2e legs, both CPU baselines), MXFP8 bench,
# parity under injected trained-tower statistics incl. the MXFP8 mode.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s5
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1800 python3 -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -15 $OUT/tests.log
timeout 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc $?"
python3 - $OUT <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], json.dumps(d["e2e"])[:1500], json.dumps(d["cpu_baseline"])[:600], d["parity"]["pass"])
PY
timeout 600 python3 bench.py --mxfp8 --steps 20 --warmup 5 > $OUT/mxfp8_bench.json 2> $OUT/mxfp8_bench.err; echo "mx rc $?"
python3 - $OUT <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/mxfp8_bench.json").read().strip().splitlines()[-1])
print("mxfp8", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k["name"]: k["ms_per_step"] for k in d["roofline"]["by_kernel"]})
PY
timeout 1500 python3 tools/parity_outliers.py --mxfp8 > $OUT/parity_outliers.json 2> $OUT/parity_outliers.err; echo "outliers rc $?"
python3 - $OUT <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/parity_outliers.json").read().strip().splitlines()[-1])
for e, v in d["encoders"].items():
    for sc, r in v.items():
        print(e, sc, {k[:-10]: (r[k]["rel_l2"], r[k]["rel_l2_bulk"]) for k in r if k.endswith("_vs_oracle")})
PY
