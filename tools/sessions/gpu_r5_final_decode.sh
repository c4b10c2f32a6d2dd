#!/bin/bash
# Round-5 closing session after the decode step's dependent-round-trip work (decode.hip only; the visual path's profile set stays): full suite
# + smoke on the final library, the same-box pair of the decode step against decode.hip as of 96df696 (ab/libmerv_hip_before_decode.so),
# and bench.py's line on the final code.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/final_decode; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log; tail -3 $O/tests.log
timeout 300 python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log; tail -2 $O/smoke.log
LIBS="merv_amd/lib/libmerv_hip.so ab/libmerv_hip_before_decode.so"
for rep in 1 2; do for lib in $LIBS; do
  echo "== rep $rep $lib"; MERV_HIP_LIB=$PWD/$lib timeout 300 python3 tools/probes/decode_kernels.py 2>/dev/null | tail -1
done; done | tee $O/decode_kernels.log
for rep in 1 2; do for lib in $LIBS; do
  MERV_HIP_LIB=$PWD/$lib timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | tail -1 > $O/bench_$(basename $lib .so)_$rep.json
  python3 - $O/bench_$(basename $lib .so)_$rep.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d.get("e2e") or {}
print(sys.argv[1].split("/")[-1], "ms_per_step", d["ms_per_step"], "e2e", e.get("generated_tok_per_s"), "prefill ms", e.get("prefill_ms"), "decode ms", e.get("decode_ms_per_token"))
PY
done; done | tee $O/bench_e2e.log
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -c 600 $O/bench.json
