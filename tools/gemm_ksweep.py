#!/usr/bin/env python3
"""Per-tile fixed cost vs per-K-tile cost of the GEMM tile configurations: exact-round shapes (M = 32768, N = 1024:
512 tiles of 256x256 = 2 rounds of 256 CUs, 1024 tiles of 256x128 = 4 rounds), K swept; fits t_round = a + b * (K/64)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os
os.environ.setdefault("MERV_TUNING_HOOKS", "1")  # forced tile configurations / kernel forms: the hooks build (merv_amd/_lib.py)
import torch

from merv_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4, 7]
M, N = 32768, 1024
rounds = {4: 4, 3: 4, 7: 2, 2: 2, 5: 2}
g = torch.Generator(device=dev).manual_seed(0)
res_flag = len(sys.argv) > 2 and sys.argv[2] == "res"
for v in variants:
    pts = []
    for K in (256, 512, 1024, 2048, 4096):
        a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=dev) * K**-0.5).to(torch.bfloat16)
        bias = torch.randn(N, generator=g, device=dev)
        r = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16) if res_flag else None
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        lib.merv_debug_set_gemm_variant(v)
        best = 1e9
        for _ in range(3):
            ops.gemm(a, w, bias=bias, res=r, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm(a, w, bias=bias, res=r, out=out)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        lib.merv_debug_set_gemm_variant(0)
        pts.append((K // 64, best / rounds[v]))
    n = len(pts)
    sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts); sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    a0 = (sy - b * sx) / n
    print(f"v{v} res={res_flag}: per-round us by K-tiles " + ", ".join(f"{k}:{t:.1f}" for k, t in pts) + f" | fit a={a0:.2f} us b={b:.3f} us/K-tile")
