#!/bin/bash
# Round-5 final GPU session: full suite + smoke, the round's profile set on the final code (tools/gpu_profile_round.sh), the same-box pair
# against the previous round's library (profiles/r05_ab_prev_round.json), the batch sweep (profiles/r05_batch_sweep.json).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1200 python3 -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -3 $OUT/tests.log
timeout 300 python3 __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
bash tools/gpu_profile_round.sh > $OUT/profile_round.log 2>&1; tail -5 $OUT/profile_round.log
# same-box pair against the previous round's library, alternating
# (round 4 = its library AND its orchestration: one stream per encoder, enqueued in index order)
for rep in 1 2 3; do
  MERV_HIP_LIB=$R/ab/libmerv_hip_r4.so MERV_ENCODER_STREAM_MAP=0123 MERV_ENCODER_ORDER=0123 timeout 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/ab_libmerv_hip_r4_$rep.json 2> $OUT/ab_libmerv_hip_r4_$rep.err
  MERV_HIP_LIB=$R/merv_amd/lib/libmerv_hip.so timeout 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/ab_libmerv_hip_$rep.json 2> $OUT/ab_libmerv_hip_$rep.err
done
python3 - $OUT <<'PY'
import json, sys
out = sys.argv[1]
runs = []
for rep in (1, 2, 3):
    for lib, tag in (("libmerv_hip_r4", "round 4: ab/libmerv_hip_r4.so (the tree at df1b89b) + its orchestration (MERV_ENCODER_STREAM_MAP=0123 MERV_ENCODER_ORDER=0123: one stream per encoder, index order)"), ("libmerv_hip", "round 5: merv_amd/lib/libmerv_hip.so + the batch-dependent stream map")):
        try:
            d = json.loads(open(f"{out}/ab_{lib}_{rep}.json").read().strip().splitlines()[-1])
            runs.append({"library": tag, "rep": rep, "tokens_per_s": d["value"], "ms_per_step": d["ms_per_step"], "gemm_roofline_frac": d["roofline"]["frac"],
                         "gemm_passes_tflops": d["roofline"].get("passes_tflops"), "e2e_gen_tok_s": d["config"]["e2e_gen_tok_s"],
                         "by_kernel_ms": {k["name"]: k["ms_per_step"] for k in d["roofline"]["by_kernel"]}})
        except Exception as e:
            runs.append({"library": tag, "rep": rep, "error": str(e)})
doc = {"what": "same-box pair: `bench.py --steps 20 --warmup 5 --no-cpu-baseline` with MERV_HIP_LIB pointing at the previous round's library and at this round's, "
               "alternating, one GPU session (boxes of the pool differ by +-3 %, so only this pair says whether a round moved the headline). Round 5 against round 4: "
               "which encoders share a stream and the order they are enqueued in (MervVisualPath.stream_map / enqueue_order); the eight-phase GEMM's MFMA issue order "
               "(2 x 2 operand blocks) and its LDS-DMA pieces in scalar-base form; the streamed attention's full key tiles as a compile-time body; the decode GEMV's "
               "fused RMSNorm as a block-level LDS image with one row x four chunks per wave (e2e leg). Same bits in every kernel.",
       "runs": runs}
json.dump(doc, open(f"{out}/ab_prev_round.json", "w"), indent=1)
print(json.dumps([(r.get("library", "")[:7], r.get("ms_per_step"), r.get("gemm_roofline_frac"), r.get("e2e_gen_tok_s")) for r in runs]))
PY
# the N > 1 code path on one rank (RCCL process group of 1 rank): plain and under torch.distributed.run as the driver launches it
MERV_BENCH_FORCE_DISTRIBUTED=1 timeout 600 python3 bench.py --steps 10 --warmup 3 --no-e2e > $OUT/forcedist_world1.json 2> $OUT/forcedist_world1.err; echo "forcedist rc $?"; grep "\[bench\]" $OUT/forcedist_world1.err | head -3
MERV_BENCH_FORCE_DISTRIBUTED=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-e2e --no-cpu-baseline > $OUT/forcedist_torchrun.json 2> $OUT/forcedist_torchrun.err; echo "torchrun rc $?"; tail -c 300 $OUT/forcedist_torchrun.json
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
               "`bench.py --batch B --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prof`; B = 24 / 32 / 48 / 64 measured 142.8 / 142.7 / 142.6 / 143.0 k against 142.1-142.4 k at 16 in another session",
       "rows": rows}
json.dump(doc, open(f"{out}/batch_sweep.json", "w"), indent=1)
print(json.dumps(doc["rows"]))
PY
