#!/usr/bin/env python3
"""Which tile configuration should take the rows the eight-phase launch leaves over: remainder-sized problems (256 rows:
LanguageBind at 16 videos; 1280: DINOv2; 3328: ViViT / SigLIP-like) on every small-tile variant, interleaved rounds, random data.
Variants: 1 = 128x128 2-stage, 4 = 256x128 staggered, 6 = 128x128 4-stage, 9 = 64x128 4 waves, 0 = what the library picks."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import os
os.environ.setdefault("MERV_TUNING_HOOKS", "1")  # forced tile configurations / kernel forms: the hooks build (merv_amd/_lib.py)
import torch

from merv_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 4, 6, 9]
cases = []
for M in (256, 1280):
    cases += [(M, 3072, 1024, "none", False), (M, 1024, 1024, "none", True), (M, 4096, 1024, "gelu_erf", False), (M, 1024, 4096, "none", True)]
for M in (3328, 6656):
    cases += [(M, 2304, 768, "none", False), (M, 768, 768, "none", True), (M, 3072, 768, "gelu_tanh", False), (M, 768, 3072, "none", True)]
g = torch.Generator(device=dev).manual_seed(0)
for M, N, K, act, res in cases:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K**-0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev)
    r = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    times = {v: [] for v in variants}
    ref = None
    for rnd in range(4):
        for v in variants:
            lib.merv_debug_set_gemm_variant(v)
            ops.gemm(a, w, bias=bias, act=act, res=r, out=out)
            if rnd == 0:
                if ref is None:
                    ref = out.clone()
                assert torch.equal(out, ref), (M, N, K, v)  # every tile configuration accumulates K in the same order: same bits
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(a, w, bias=bias, act=act, res=r, out=out)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 20)
    lib.merv_debug_set_gemm_variant(0)
    print(f"M={M:5d} N={N:5d} K={K:5d} " + " | ".join(f"v{v}: {min(t)*1e3:6.1f} us" for v, t in times.items()), flush=True)
