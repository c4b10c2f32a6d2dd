#!/usr/bin/env python3
"""prefill_breakdown.py [S]: where the prefill of the e2e leg goes (Llama-2-7B geometry, random init, S = 1049 tokens):
wall time of `dec.prefill`, and the per-kernel GPU time of one prefill from torch.profiler (kernel name, calls, total us)."""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from merv_amd.llm import LlamaBackbone, HipDecoder, llama2_7b_config

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1049
dev = torch.device("cuda:0")
llm = LlamaBackbone(llama2_7b_config(), device=dev)
with torch.inference_mode():
    dec = HipDecoder(llm.llm, 1280, 1)
    emb = torch.randn(1, S, llm.config.hidden_size, device=dev, dtype=torch.bfloat16) * 0.02
    for _ in range(2): dec.prefill(emb)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); dec.prefill(emb); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("prefill wall ms", round(best * 1e3, 2), "S", S, "variant", getattr(dec, "prefill_variant", "torch"))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        dec.prefill(emb); torch.cuda.synchronize()
    rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0 and e.device_type.name != "CPU"]
    if not rows:
        rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0]
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    print("kernels: GPU us total", round(tot, 1))
    for k, c, t in rows[:40]:
        print("%9.1f us %5d  %s" % (t, c, k[:110]))
