#!/usr/bin/env python3
"""sdpa_probe.py: PyTorch-ROCm SDPA on the prefill shape (S = 1049, 32 heads, head dim 128, causal) by backend and layout."""
import torch, time
import torch.nn.functional as F
from torch.nn.attention import sdpa_kernel, SDPBackend
dev = torch.device("cuda:0")
S, H, hd = 1049, 32, 128
g = torch.Generator(device=dev).manual_seed(0)
qs = torch.randn(1, S, H, hd, device=dev, dtype=torch.bfloat16, generator=g)
ks, vs = torch.randn_like(qs), torch.randn_like(qs)
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
lay = {"views of [S,H,hd]": (qs.transpose(1, 2), ks.transpose(1, 2), vs.transpose(1, 2)),
       "contiguous [H,S,hd]": (qs.transpose(1, 2).contiguous(), ks.transpose(1, 2).contiguous(), vs.transpose(1, 2).contiguous())}
for ln, (q, k, v) in lay.items():
    for bn, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
        try:
            with sdpa_kernel(be):
                t = bench(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=True))
            print(f"{ln:22s} {bn:10s} {t:8.1f} us")
        except Exception as e:
            print(f"{ln:22s} {bn:10s} failed: {str(e)[:80]}")
t = bench(lambda: F.scaled_dot_product_attention(*lay["views of [S,H,hd]"], is_causal=True))
print("default dispatch", round(t, 1), "us")
