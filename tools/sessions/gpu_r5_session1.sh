#!/bin/bash
# Round-5 GPU session 1 (via gpurun): test suite on the refactored library, the operand-wrap energy bound of the eight-phase GEMM with
# in-kernel clocks, the batch sweep BASELINE.md config 2 asks for, the N > 1 code path on one rank, decode per-launch times.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s1
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -4 $OUT/tests.log
timeout 600 ./build/gemm_energy_bound 7 0.3 > $OUT/gemm_energy_bound.json 2> $OUT/gemm_energy_bound.err; echo "energy bound rc $?"; head -c 1500 $OUT/gemm_energy_bound.json
for B in 1 2 4 8 16; do
  timeout 300 python3 bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prof > $OUT/sweep_b$B.json 2> $OUT/sweep_b$B.err
done
python3 - $OUT <<'PY'
import json, sys
out = sys.argv[1]
rows = []
for B in (1, 2, 4, 8, 16):
    try:
        d = json.loads(open(f"{out}/sweep_b{B}.json").read().strip().splitlines()[-1])
        rows.append({"videos_per_step": B, "tokens_per_s": d["value"], "ms_per_step": d["ms_per_step"], "path_tflops": d["config"]["path_tflops"],
                     "path_frac_of_mfma_peak": d["config"]["path_frac_of_mfma_peak"]})
    except Exception as e:
        rows.append({"videos_per_step": B, "error": str(e)})
doc = {"what": "batch sweep of bench.py (BASELINE.md config 2): fused visual tokens/s at B videos per step, one box, one session, concurrent encoder streams, "
               "`bench.py --batch B --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prof`", "rows": rows}
json.dump(doc, open(f"{out}/batch_sweep.json", "w"), indent=1)
print(json.dumps(doc))
PY
MERV_BENCH_FORCE_DISTRIBUTED=1 timeout 600 python3 bench.py --steps 10 --warmup 3 --no-e2e > $OUT/forcedist_world1.json 2> $OUT/forcedist_world1.err; echo "forcedist rc $?"; head -c 600 $OUT/forcedist_world1.json; grep "\[bench\]" $OUT/forcedist_world1.err
timeout 300 python3 tools/probes/decode_kernels.py > $OUT/decode_kernels.json 2> $OUT/decode_kernels.err; cat $OUT/decode_kernels.json
MERV_HIP_LIB=$R/ab/libmerv_hip_r4.so timeout 300 python3 tools/probes/decode_kernels.py > $OUT/decode_kernels_r4.json 2> $OUT/decode_kernels_r4.err; cat $OUT/decode_kernels_r4.json
