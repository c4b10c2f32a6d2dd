#!/usr/bin/env python3
"""prefill_gemm_m_sweep.py: library GEMM time of the prefill's projections against the row count around the e2e prompt's 1049 positions
(padding to 1152 rows would make the fused q / k / v GEMM 9 % faster, the others 0-2 %: 0.3 ms of a 16 ms prefill -- not done)."""
import torch, time
import torch.nn.functional as F
dev = torch.device("cuda:0")
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.02).to(torch.bfloat16)
for name, N, K in (("qkv fused", 12288, 4096), ("o", 4096, 4096), ("gate", 11008, 4096), ("down", 4096, 11008)):
    W = mk(N, K)
    row = []
    for M in (1024, 1049, 1056, 1152, 1280, 25):
        x = mk(M, K)
        row.append("M=%d: %.1f us" % (M, bench(lambda: F.linear(x, W))))
    print(name, " | ".join(row))
