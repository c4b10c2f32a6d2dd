#!/bin/bash
# Round-6 final GPU session: full suite + smoke, the round's profile set on the final code (tools/gpu_profile_round.sh), the same-box pair against the
# previous round's library at 16 videos (profiles/r06_ab_prev_round.json), the batch sweep beside the round-5 library (profiles/r06_batch_sweep.json),
# the MXFP8 bench line, the self-spawned distributed leg.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1800 python3 -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "pytest rc $?" >> $OUT/tests.log; tail -3 $OUT/tests.log
timeout 300 python3 __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
bash tools/gpu_profile_round.sh > $OUT/profile_round.log 2>&1; tail -3 $OUT/profile_round.log | cut -c1-300
export MERV_HIP_LIB_AB=1
for rep in 1 2 3; do
  MERV_HIP_LIB=$R/ab/libmerv_hip_r5.so timeout 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/ab_libmerv_hip_r5_$rep.json 2> $OUT/ab_libmerv_hip_r5_$rep.err
  MERV_HIP_LIB=$R/merv_amd/lib/libmerv_hip.so timeout 400 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/ab_libmerv_hip_$rep.json 2> $OUT/ab_libmerv_hip_$rep.err
done
for lib in ab/libmerv_hip_r5.so merv_amd/lib/libmerv_hip.so; do for B in 1 2 4 8 16; do
  MERV_HIP_LIB=$R/$lib timeout 300 python3 bench.py --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-prof 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib B $B ms_per_step', d['ms_per_step'], 'tokens/s', d['value'], 'frac', d['config']['path_frac_of_mfma_peak'])
" | tee -a $OUT/batch_sweep.txt
done; done
unset MERV_HIP_LIB_AB
python3 - $OUT <<'PY'
import json, sys
out = sys.argv[1]
runs = []
for rep in (1, 2, 3):
    for lib, tag in (("libmerv_hip_r5", "round 5: ab/libmerv_hip_r5.so (the tree at 4316074), this round's Python side"), ("libmerv_hip", "round 6: merv_amd/lib/libmerv_hip.so")):
        try:
            d = json.loads(open(f"{out}/ab_{lib}_{rep}.json").read().strip().splitlines()[-1])
            runs.append({"library": tag, "rep": rep, "tokens_per_s": d["value"], "ms_per_step": d["ms_per_step"], "gemm_roofline_frac": d["roofline"]["frac"],
                         "e2e_gen_tok_s": d["config"]["e2e_gen_tok_s"], "e2e_quick_start_sampled_tok_s": d["config"].get("e2e_gen_tok_s_quick_start_sampled_512"),
                         "visual_path_ms": (d.get("e2e") or {}).get("visual_path_ms"), "decode_ms_per_token": (d.get("e2e") or {}).get("decode_ms_per_token")})
        except Exception as e:
            runs.append({"library": tag, "rep": rep, "error": str(e)})
doc = {"what": "same-box pair: `bench.py --steps 20 --warmup 5 --no-cpu-baseline` with MERV_HIP_LIB pointing at the previous round's library and at this round's, alternating, one GPU "
               "session. The 16-video bf16 kernels did not change in round 6 (same bits; the M0 clobber on the GEMM's DMA statements and the hooks compiled out are the only differences "
               "in that path), so the headline pair is expected to tie; the e2e legs differ through the one-video visual path (sub-round tile threshold, enqueue order). The round-5 "
               "library has no merv_decode_sample_advance: its sampled leg falls back to the host loop.", "runs": runs}
json.dump(doc, open(f"{out}/ab_prev_round.json", "w"), indent=1)
print(json.dumps([(r.get("library", "")[:7], r.get("ms_per_step"), r.get("gemm_roofline_frac"), r.get("e2e_gen_tok_s"), r.get("e2e_quick_start_sampled_tok_s"), r.get("visual_path_ms")) for r in runs]))
PY
timeout 600 python3 bench.py --mxfp8 --steps 20 --warmup 5 > $OUT/mxfp8_bench.json 2> $OUT/mxfp8_bench.err; echo "mx rc $?"
MERV_BENCH_FORCE_DISTRIBUTED=1 timeout 900 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-e2e > $OUT/forcedist_selfspawn.json 2> $OUT/forcedist_selfspawn.err; echo "forcedist rc $?"; grep "\[bench\]" $OUT/forcedist_selfspawn.err | head -3 | cut -c1-250
