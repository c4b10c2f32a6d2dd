#!/bin/bash
# MFMA utilisation, instruction mix, L2 hit rate and HBM traffic PER GEMM CLASS of the bench step (16 videos, encoders one after
# another): four separate rocprofv3 --pmc passes (--kernel-trace only, the program directly after --) over
#   bench.py --sequential --steps 1 --warmup 1 --no-prof --no-cpu-baseline --no-e2e        (2 identical steps)
# -> gpurun_out/pmc_round/pmc_gemm.json (copy to profiles/rNN_pmc_gemm.json)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_round
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --sequential --steps 1 --warmup 1 --no-prof --no-cpu-baseline --no-e2e"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $B > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/b -- python3 $B > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c -- python3 $B > $OUT/c.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/d -- python3 $B > $OUT/d.log 2>&1
python3 - $OUT <<'PY'
import csv, sys, glob, json, re, collections
out = sys.argv[1]
STEPS = 2
CLASSES = [
    (r"gemm_bf16_8phase_kernel<false, 0, false, 3(, \d+)?>", "eight-phase, folded LayerNorm (qkv)"),
    (r"gemm_bf16_8phase_kernel<false, 0, false, 1(, \d+)?>", "eight-phase, bias only (proj / fc2 / temporal qkv, proj / projector; ViViT, SigLIP, LanguageBind)"),
    (r"gemm_bf16_8phase_kernel<false, 0, false, 2(, \d+)?>", "eight-phase, LayerScale (DINOv2 proj / fc2)"),
    (r"gemm_bf16_8phase_kernel<false, [123], false, \d(, \d+)?>", "eight-phase + activation, direct epilogue (fc1)"),
    (r"gemm_bf16_kernel<", "small tiles"),
    (r"attn_kernel<true, 4, 2, true, true(, \w+)?>", "attention, K/V resident"),
    (r"(?<!temporal_)attn_kernel<", "attention, K/V streamed"),
    (r"temporal_attn_kernel", "temporal attention"),
]
def cls(name):
    for pat, c in CLASSES:
        if re.search(pat, name): return c
    return None
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for d in "abcd":
    fs = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        c = cls(r["Kernel_Name"])
        if c is None: continue
        agg[c][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(c, d)].add(r["Dispatch_Id"])
dur = collections.defaultdict(float); ndur = collections.defaultdict(int)
fs = glob.glob(f"{out}/c/**/*kernel_trace.csv", recursive=True)   # durations from the lightest pass
if fs:
    for r in csv.DictReader(open(fs[0])):
        c = cls(r["Kernel_Name"])
        if c is None: continue
        dur[c] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])); ndur[c] += 1
PASS = {"a": ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"),
        "c": ("FETCH_SIZE",), "d": ("WRITE_SIZE",)}
res = {}
for c, cnt in agg.items():
    n = {d: max(1, len(disp[(c, d)])) for d in "abcd"}
    per = {}
    for k, v in cnt.items():
        d = next((p for p, ks in PASS.items() if k in ks), "b")
        per[k] = v / n[d]
    der = {}
    if per.get("GRBM_GUI_ACTIVE"):
        der["mfma_busy_frac (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs))"] = round(per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * per["GRBM_GUI_ACTIVE"] / 8), 4)
    if per.get("SQ_WAVE_CYCLES"):
        w = per["SQ_WAVE_CYCLES"]
        der["wave_time_waiting_frac (s_waitcnt / barrier)"] = round(per.get("SQ_WAIT_ANY", 0) / w, 4)
        der["wave_time_issue_stalled_frac"] = round(per.get("SQ_WAIT_INST_ANY", 0) / w, 4)
    if per.get("SQ_INSTS_MFMA"):
        der["valu_insts_per_mfma"] = round(per.get("SQ_INSTS_VALU", 0) / per["SQ_INSTS_MFMA"], 2)
    if per.get("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_frac"] = round(per.get("SQ_LDS_BANK_CONFLICT", 0) / per["SQ_LDS_IDX_ACTIVE"], 4)
    if per.get("TCC_HIT_sum") is not None and (per.get("TCC_HIT_sum", 0) + per.get("TCC_MISS_sum", 0)) > 0:
        der["l2_hit_rate"] = round(per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"]), 4)
    if "FETCH_SIZE" in per or "WRITE_SIZE" in per:
        hb = (2.0 * per.get("FETCH_SIZE", 0) + per.get("WRITE_SIZE", 0)) * 1024.0   # gfx950: FETCH_SIZE counts 128-B requests at 64 B
        der["hbm_bytes_per_dispatch (2 x FETCH_SIZE + WRITE_SIZE)"] = round(hb)
        der["hbm_gb_per_step"] = round(hb * n["c"] / STEPS / 1e9, 3)
        if ndur[c]:
            der["hbm_tb_per_s_while_running"] = round(hb / (dur[c] / ndur[c]) / 1e3, 3)
    res[c] = {"dispatches_per_step": n["a"] / STEPS, "avg_us (trace of the FETCH_SIZE pass)": round(dur[c] / max(1, ndur[c]) / 1e3, 2),
              "ms_per_step": round(dur[c] / STEPS / 1e6, 3), "per_dispatch": {k: round(v, 1) for k, v in sorted(per.items())}, "derived": der}
doc = {"what": "tools/pmc_gemm_round.sh: rocprofv3 --pmc in four separate passes (SQ + GRBM; SQ instruction counts + TCC hit / miss; FETCH_SIZE; "
               "WRITE_SIZE), --kernel-trace only, over `bench.py --sequential --steps 1 --warmup 1 --no-prof` (2 steps of 16 videos, encoders "
               "one after another); per kernel class, per-dispatch averages. SQ_VALU_MFMA_BUSY_CYCLES counts cycles (16 per 16x16x32 bf16 MFMA, "
               "32 per 32x32x16), GRBM_GUI_ACTIVE is summed over the 8 XCDs.", "steps_in_run": STEPS, "classes": res}
json.dump(doc, open(f"{out}/pmc_gemm.json", "w"), indent=1)
for c, v in res.items():
    print(c, v["dispatches_per_step"], v["ms_per_step"], json.dumps(v["derived"]))
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
