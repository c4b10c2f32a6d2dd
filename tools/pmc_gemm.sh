#!/bin/bash
# PMC counters for the eight-phase GEMM (bf16 and MXFP8) on the encoder shapes, own runs with --kernel-trace only.
# Output: gpurun_out/pmc/gemm_pmc.json (per kernel: counters per dispatch, derived fractions).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -- python3 $R/tools/mx_gemm_bench.py > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 $R/tools/mx_gemm_bench.py > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/c -- python3 $R/tools/mx_gemm_bench.py > $OUT/c.log 2>&1
tail -2 $OUT/a.log
python3 - $OUT <<'PY'
import csv, sys, glob, json, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set); dur = collections.defaultdict(float)
for d in "abc":
    fs = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "8phase" not in k: continue
        name = "mxfp8" if "true>" in k.replace(" ", "") and k.rstrip().endswith("true>(merv::GemmArgs)") else "bf16"
        name = "eight-phase " + ("MXFP8" if ", true>(" in k else "bf16")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(name, d)].add(r["Dispatch_Id"])
res = {}
for name, c in agg.items():
    n = {d: max(1, len(disp[(name, d)])) for d in "abc"}
    per = {}
    for k, v in c.items():
        d = "a" if k in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS") else ("c" if k.startswith("TCC") else "b")
        per[k] = v / n[d]
    der = {}
    if per.get("SQ_WAVE_CYCLES"):
        w = per["SQ_WAVE_CYCLES"]
        der["wave_time_waiting_frac (s_waitcnt / barrier)"] = per.get("SQ_WAIT_ANY", 0) / w
        der["wave_time_issue_stalled_frac"] = per.get("SQ_WAIT_INST_ANY", 0) / w
        der["wave_time_issuing_frac"] = per.get("SQ_ACTIVE_INST_ANY", 0) / w
    if per.get("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_frac"] = per.get("SQ_LDS_BANK_CONFLICT", 0) / per["SQ_LDS_IDX_ACTIVE"]
    if per.get("TCC_HIT_sum") is not None and per.get("TCC_MISS_sum") is not None and (per["TCC_HIT_sum"] + per["TCC_MISS_sum"]) > 0:
        der["l2_hit_rate"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
    if per.get("SQ_INSTS_MFMA") and per.get("SQ_INSTS_VALU"):
        der["valu_insts_per_mfma"] = per["SQ_INSTS_VALU"] / per["SQ_INSTS_MFMA"]
    res[name] = {"dispatches_sampled": n, "per_dispatch": per, "derived": der}
doc = {"note": "rocprofv3 --pmc, three separate passes over tools/mx_gemm_bench.py (encoder GEMM shapes at B=8, bf16 and MXFP8 "
               "launches of the eight-phase kernel, all shapes pooled); SQ_*CYCLES in the units the tool reports", "kernels": res}
json.dump(doc, open(f"{out}/gemm_pmc.json", "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in res.items()}, indent=1))
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
