#!/usr/bin/env python3
"""Attention / LayerNorm / temporal-attention micro-benchmark on the encoder stack's real shapes (GPU box)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from merv_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, n=10, rounds=3):
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


for name, nseq, L, heads in [("languagebind", 16 * B, 257, 16), ("dinov2", 16 * B, 261, 16), ("siglip", 16 * B, 196, 12),
                             ("vivit", B, 3137, 12)]:
    D = heads * 64
    qkv = (torch.randn(nseq * L, 3 * D, generator=g, device=dev) * 1.5).to(torch.bfloat16)
    out = ops.attention(qkv, nseq, L, heads)
    q, k, v = qkv[: 2 * L].float().reshape(2, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = ((q @ k.transpose(-1, -2) * 0.125).softmax(-1) @ v).transpose(1, 2).reshape(2 * L, D) if name != "vivit" else None
    if ref is not None:
        err = float((out[: 2 * L].float() - ref).norm() / ref.norm())
        assert err < 1e-2 or os.environ.get("ATTN_BENCH_NOCHECK") == "1", err  # (ablation libraries compute garbage)
    t = timeit(lambda: ops.attention(qkv, nseq, L, heads))
    fl = 4.0 * nseq * L * L * D
    print(f"attn {name:13s} nseq={nseq:4d} L={L:5d} heads={heads}: {t*1e3:8.1f} us  {fl/t/1e9:7.1f} TF", flush=True)

M, D = 4112 * B, 1024
x = torch.randn(M, D, generator=g, device=dev).to(torch.bfloat16)
gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
t = timeit(lambda: ops.layernorm(x, gam, bet, 1e-6))
print(f"layernorm M={M} D={D}: {t*1e3:.1f} us  {M*D*4/t/1e6:.0f} GB/s (read+write bf16)")
qkv = torch.randn(M, 3 * D, generator=g, device=dev).to(torch.bfloat16)
t = timeit(lambda: ops.temporal_attention(qkv, 2 * B, 8, 257, 16))
print(f"temporal attn rows={M}: {t*1e3:.1f} us  {M*D*2*4/t/1e6:.0f} GB/s (q,k,v read + out write)")
