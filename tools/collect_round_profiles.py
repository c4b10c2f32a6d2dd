#!/usr/bin/env python3
"""collect_round_profiles.py <round tag, e.g. r04>: copies what tools/gpu_profile_round.sh left under gpurun_out/round/ (and the
attention PMC log under gpurun_out/round/pmc_attn.log) into profiles/<tag>_*: bench line, rocprofv3 kernel stats, the per-class
roofline joined with the PMC busy fractions of the GEMM / attention classes, the per-class PMC file, fabric traffic, attention PMC."""
import ast, json, re, shutil, sys
from pathlib import Path

tag = sys.argv[1]
root = Path(__file__).resolve().parent.parent
src, dst = root / "gpurun_out" / "round", root / "profiles"
shutil.copy(src / "bench.json", dst / f"{tag}_bench.json")
shutil.copy(src / "trace_bench.json", dst / f"{tag}_bench_under_rocprof_sequential.json")
shutil.copy(src / "kernel_stats.csv", dst / f"{tag}_kernel_stats.csv")
shutil.copy(src / "pmc_gemm.json", dst / f"{tag}_pmc_gemm.json")
shutil.copy(src / "pmc_gemm_traffic.json", dst / f"{tag}_pmc_gemm_traffic.json")

roof = json.load(open(src / "kernel_roofline.json"))
pmc = json.load(open(src / "pmc_gemm.json"))["classes"]
CLASS_OF = [("eight-phase + activation", "gemm eight-phase + activation epilogue"), ("eight-phase", "gemm eight-phase, no activation"),
            ("small tiles", "gemm small tiles"), ("attention, K/V resident", "attention, K/V resident"),
            ("attention, K/V streamed", "attention, K/V streamed"), ("temporal attention", "temporal attention")]
key = f"pmc (profiles/{tag}_pmc_gemm.json, separate rocprofv3 --pmc passes of the same step)"
for c in roof["classes"]:
    c[key] = {}
for name, v in pmc.items():
    target = next(t for p, t in CLASS_OF if name.startswith(p))
    der = v.get("derived", {})
    pick = lambda s: next((val for k, val in der.items() if k.startswith(s)), None)
    entry = {"ms_per_step": v.get("ms_per_step"), "mfma_busy_frac": pick("mfma_busy_frac"), "valu_insts_per_mfma": pick("valu_insts_per_mfma"),
             "l2_hit_rate": pick("l2_hit_rate"), "hbm_gb_per_step": pick("hbm_gb_per_step")}
    next(c for c in roof["classes"] if c["name"] == target)[key][name] = entry
roof["classes"] = [{k: v for k, v in c.items() if not (k == key and not v)} for c in roof["classes"]]
json.dump(roof, open(dst / f"{tag}_kernel_roofline.json", "w"), indent=1)

# attention PMC: the two passes' per-kernel lines of tools/pmc_attn.sh
kern = {}
for line in open(src / "pmc_attn.log"):
    m = re.match(r"\('(.*?)', '(\d+)'\) dispatches (\d+) (\{.*\})", line.strip())
    if not m:
        continue
    k = f"...{m.group(1)} grid={m.group(2)}"
    kern.setdefault(k, {"dispatches": int(m.group(3))}).update({a: float(b) for a, b in ast.literal_eval(m.group(4)).items()})
for k, v in kern.items():
    if v.get("SQ_INSTS_MFMA"):
        v["valu_insts_per_mfma"] = round(v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"], 2)
    if v.get("SQ_WAVE_CYCLES"):
        v["wave_time_waiting_frac"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3)
json.dump({"what": "tools/pmc_attn.sh: rocprofv3 --pmc (two separate passes, --kernel-trace only) over tools/attn_bench.py 8; per-dispatch averages. "
                   "grid=524288: the resident 4x2 kernel on 257 / 261-token sequences (LanguageBind, DINOv2); 393216: 196 tokens (SigLIP); "
                   "319488: 3137 tokens (ViViT); 1052672: temporal attention.", "kernels": kern}, open(dst / f"{tag}_pmc_attn.json", "w"), indent=1)
print("wrote", sorted(p.name for p in dst.glob(f"{tag}_*")))
