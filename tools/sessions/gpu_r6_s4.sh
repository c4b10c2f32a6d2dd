#!/bin/bash
# Round 6, session 4: MXFP8 mode under the LayerNorm fold (static epilogue forms, the stream's MXFP8 copy from the producing epilogue, resident
# attention with MXFP8 output): suite, bench --mxfp8 beside bf16 on one box, accuracy per mask; batch sweep against the round-5 library.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s4
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -12 $OUT/tests.log
for rep in 1 2; do
  timeout 600 python3 bench.py --mxfp8 --steps 20 --warmup 5 > $OUT/mxfp8_bench_$rep.json 2> $OUT/mxfp8_bench_$rep.err; echo "mx rc $?"
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $OUT/bf16_bench_$rep.json 2> $OUT/bf16_bench_$rep.err; echo "bf16 rc $?"
  python3 - $OUT $rep <<'PY'
import json, sys
out, rep = sys.argv[1], sys.argv[2]
for tag in ("mxfp8", "bf16"):
    try:
        d = json.loads(open(f"{out}/{tag}_bench_{rep}.json").read().strip().splitlines()[-1])
        print(tag, rep, d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], {k["name"]: k["ms_per_step"] for k in d["roofline"]["by_kernel"]})
    except Exception as e:
        print(tag, rep, "failed", e)
PY
done
timeout 900 python3 tools/mx_accuracy.py > $OUT/mx_accuracy.log 2>&1; cp gpurun_out/mx_accuracy.json $OUT/ 2>/dev/null; tail -8 $OUT/mx_accuracy.log
export MERV_HIP_LIB_AB=1
for lib in ab/libmerv_hip_r5.so merv_amd/lib/libmerv_hip.so; do for B in 1 2 4 8 16; do
  MERV_HIP_LIB=$R/$lib timeout 300 python3 bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib B $B ms_per_step', d['ms_per_step'], 'tokens/s', d['value'], 'frac', d['config']['path_frac_of_mfma_peak'])
" | tee -a $OUT/batch_sweep.txt
done; done
