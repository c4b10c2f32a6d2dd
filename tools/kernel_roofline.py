#!/usr/bin/env python3
"""Joins bench.py's `roofline.by_kernel` (algorithmic FLOPs / bytes per step and HIP-event time per kernel class) with a
rocprofv3 --kernel-trace --stats summary of the same workload, so every per-kernel roofline fraction can be re-divided from
the profiler's own durations:

    python3 tools/kernel_roofline.py <bench.json> <kernel_stats.csv> <steps in the traced run> <out.json>

The traced run must be `bench.py --sequential --no-prof --no-cpu-baseline --no-e2e --steps K --warmup W` (steps = K + W: every
step launches the same kernels) at the same --batch as <bench.json>."""
import csv
import json
import re
import sys

# kernel-name pattern -> by_kernel class name (bench.py KERNEL_CLASSES)
RULES = [
    (r"gemm_bf16_8phase_kernel<false, 0, false(, \d+)*>", "gemm eight-phase, no activation"),   # <REMAP, ACT, MX(, EPI: round 4)(, WHOLE: round 5)>
    (r"gemm_bf16_8phase_kernel<false, [123], false(, \d+)*>", "gemm eight-phase + activation epilogue"),
    (r"gemm_bf16_kernel<", "gemm small tiles"),
    (r"attn_kernel<true, 4, 2, true, true(, \w+)?>", "attention, K/V resident"),
    (r"(?<!temporal_)attn_kernel<", "attention, K/V streamed"),
    (r"temporal_attn_kernel", "temporal attention"),
    (r"layernorm_kernel", "LayerNorm"),
    (r"row_stats_kernel|stats_finalize_kernel", "LayerNorm statistics"),
    (r"pool_kernel|fusion_score_kernel|fusion_mix_kernel", "pool + fusion"),
    (r"im2col_kernel|im2col_fast_kernel|prefix_kernel|gather_tokens_kernel", "data movement"),
]


def main(bench_json, stats_csv, steps, out):
    steps = int(steps)
    line = json.loads(open(bench_json).read().strip().splitlines()[-1])
    by = {e["name"]: e for e in line["roofline"]["by_kernel"]}
    trace = {}
    for r in csv.DictReader(open(stats_csv)):
        if "merv::" not in r["Name"]:
            continue
        for pat, cls in RULES:
            if re.search(pat, r["Name"]):
                t = trace.setdefault(cls, {"ns": 0.0, "calls": 0, "kernels": []})
                t["ns"] += float(r["TotalDurationNs"]); t["calls"] += int(r["Calls"])
                t["kernels"].append({"name": re.sub(r"^void |merv::\(anonymous namespace\)::|merv::|\(merv::\w+\)$", "", r["Name"]),
                                     "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)})
                break
    rows = []
    for name, e in by.items():
        t = trace.get(name)
        row = {"name": name, "bound": e["bound"], "unit": e["unit"], "peak": e["peak"],
               "algorithmic_per_step": e.get("algorithmic_tflop_per_step", e.get("algorithmic_gb_per_step")),
               "algorithmic_unit": "TFLOP" if e["bound"] == "mfma" else "GB",
               "events": {"launches_per_step": e["launches_per_step"], "ms_per_step": e["ms_per_step"], "achieved": e["achieved"], "frac": e["frac"]}}
        if t:
            ms = t["ns"] / 1e6 / steps
            ach = row["algorithmic_per_step"] / (ms * 1e-3) * (1.0 if e["bound"] == "mfma" else 1.0)
            row["rocprofv3"] = {"launches_per_step": round(t["calls"] / steps, 1), "ms_per_step": round(ms, 3), "achieved": round(ach, 1),
                                "frac": round(ach / e["peak"], 4), "kernels": sorted(t["kernels"], key=lambda k: -k["calls"] * k["avg_us"])}
        rows.append(row)
    gemm = [r for r in rows if r["name"].startswith("gemm") and "rocprofv3" in r]
    doc = {"what": "per-kernel-class roofline: algorithmic work per step (bench.py) / kernel time per step, by HIP events (bench.py's second "
                   "profiler pass) and by rocprofv3 --kernel-trace --stats of `bench.py --sequential --no-prof` (this file's csv sibling)",
           "videos_per_step": line["config"]["videos_per_gpu_per_step"], "steps_in_trace": steps,
           "peaks": {"mfma_bf16_dense_TFLOPs": 2500.0, "hbm_GBs": 8000.0}, "classes": rows}
    if gemm:
        fl = sum(r["algorithmic_per_step"] for r in gemm); ms = sum(r["rocprofv3"]["ms_per_step"] for r in gemm)
        doc["gemm_family_rocprofv3"] = {"tflop_per_step": round(fl, 3), "ms_per_step": round(ms, 3), "achieved_TFLOPs": round(fl / ms * 1e3, 1),
                                        "frac": round(fl / ms * 1e3 / 2500.0, 4)}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in doc.items() if k != "classes"}))
    for r in rows:
        print(f'{r["name"]:42s} events {r["events"]["ms_per_step"]:8.3f} ms {r["events"]["frac"]:.3f}'
              + (f'   rocprof {r["rocprofv3"]["ms_per_step"]:8.3f} ms {r["rocprofv3"]["frac"]:.3f}' if "rocprofv3" in r else ""))


if __name__ == "__main__":
    main(*sys.argv[1:5])
