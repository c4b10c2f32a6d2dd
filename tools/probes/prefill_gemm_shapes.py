#!/usr/bin/env python3
"""prefill_gemm_shapes.py: library GEMM times on the prefill shapes (M = 1049), separate projections vs concatenated weights."""
import torch, time
import torch.nn.functional as F
dev = torch.device("cuda:0")
M = 1049
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.02).to(torch.bfloat16)
x, xi = mk(M, 4096), mk(M, 11008)
for name, N, K, inp in (("q/k/v/o", 4096, 4096, x), ("qkv fused", 12288, 4096, x), ("gate or up", 11008, 4096, x), ("gate+up fused", 22016, 4096, x), ("down", 4096, 11008, xi)):
    W = mk(N, K)
    t = bench(lambda: F.linear(inp, W))
    print(f"{name:14s} N={N:6d} K={K:6d}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s")
