#!/usr/bin/env python3
"""generate_overhead.py: what a generated token costs outside the decode step itself (Llama-2-7B geometry, 1049 prompt positions):
GPU time of the PyTorch kernels around the replayed step (argmax, token copy, position increment ...) and the wall time per token."""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from merv_amd.llm import LlamaBackbone, llama2_7b_config
dev = torch.device("cuda:0")
llm = LlamaBackbone(llama2_7b_config(), device=dev)
llm.config.eos_token_id = None
emb = torch.randn(1, 1049, 4096, device=dev, dtype=torch.bfloat16) * 0.02
for _ in range(2): llm.generate_from_embeds(emb, max_new_tokens=64)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); llm.generate_from_embeds(emb, max_new_tokens=64); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("generate_from_embeds(64 tokens): %.2f ms" % (min(ts) * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    llm.generate_from_embeds(emb, max_new_tokens=64); torch.cuda.synchronize()
rows = sorted(((e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0 and ("at::" in e.key or "Memcpy" in e.key or "Memset" in e.key or "elementwise" in e.key or "reduce" in e.key)), key=lambda r: -r[2])
for k, c, t in rows[:14]: print("%9.1f us %5d  %s" % (t, c, k[:110]))
