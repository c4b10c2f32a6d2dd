#!/usr/bin/env python3
"""GEMM micro-benchmark on the GPU box: times every tile configuration of merv_gemm_bf16 on the encoder stack's real
shapes (random data, interleaved rounds in one process) and checks each against torch fp32."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
os.environ.setdefault("MERV_TUNING_HOOKS", "1")  # forced tile configurations / kernel forms: the hooks build (merv_amd/_lib.py)
import torch

from merv_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3]
M_lb, M_vv = 4112 * B, 3137 * B
shapes = [
    ("lb.qkv", M_lb, 3072, 1024, "none", False), ("lb.proj", M_lb, 1024, 1024, "none", True),
    ("lb.fc1", M_lb, 4096, 1024, "gelu_erf", False), ("lb.fc2", M_lb, 1024, 4096, "none", True),
    ("vv.qkv", M_vv, 2304, 768, "none", False), ("vv.proj", M_vv, 768, 768, "none", True),
    ("vv.fc1", M_vv, 3072, 768, "gelu_tanh", False), ("vv.fc2", M_vv, 768, 3072, "none", True),
    ("lb.embed", 4096 * B, 1024, 640, "none", False), ("projector", 1024 * B, 4096, 1024, "none", False),
]
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, K, act, res in shapes:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K**-0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev)
    r = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if res and os.environ.get("GEMM_BENCH_INPLACE") == "1":  # x += linear(..): what the encoder's proj / fc2 launches do
        out = r
    ref = None
    line = f"{name:10s} M={M:6d} N={N:5d} K={K:5d}"
    times = {v: [] for v in variants}
    for rnd in range(3):
        for v in variants:
            lib.merv_debug_set_gemm_variant(v)
            ops.gemm(a, w, bias=bias, act=act, res=r, out=out)
            if rnd == 0:
                if ref is None:
                    y = a.float() @ w.float().t() + bias
                    if act == "gelu_erf":
                        y = torch.nn.functional.gelu(y)
                    elif act == "gelu_tanh":
                        y = 0.5 * y * (1 + torch.tanh(y * 0.7978845608 * (1 + 0.044715 * y * y)))
                    ref = y + (r.float() if res else 0)
                err = float((out.float() - ref).norm() / ref.norm())
                assert out is r or err < 6e-3, (name, v, err)
            n = 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.gemm(a, w, bias=bias, act=act, res=r, out=out)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / n)
    for v in variants:
        if times[v]:
            t = min(times[v])
            line += f" | v{v}: {t*1e3:7.1f} us {2.0*M*N*K/t/1e9:7.1f} TF"
    print(line, flush=True)
lib.merv_debug_set_gemm_variant(0)
