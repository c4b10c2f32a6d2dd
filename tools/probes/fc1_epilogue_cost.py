#!/usr/bin/env python3
"""What the fc1 epilogue's activation costs: the LanguageBind fc1 shape at 16 videos with every activation kind, interleaved
rounds in one process (random data)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch

from merv_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, K in (("lb.fc1", 4112 * B, 4096, 1024), ("vv.fc1", 3137 * B, 3072, 768)):
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K**-0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    acts = ["none", "gelu_erf", "gelu_tanh", "quick_gelu"]
    times = {k: [] for k in acts}
    for rnd in range(4):
        for act in acts:
            ops.gemm(a, w, bias=bias, act=act, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm(a, w, bias=bias, act=act, out=out)
            e1.record()
            torch.cuda.synchronize()
            times[act].append(e0.elapsed_time(e1) / 10)
    print(name, " | ".join(f"{k}: {min(v)*1e3:7.1f} us {2.0*M*N*K/min(v)/1e9:6.0f} TF" for k, v in times.items()), flush=True)
