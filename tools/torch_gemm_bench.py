#!/usr/bin/env python3
"""Vendor-library reference point (measurement only): torch F.linear (hipBLASLt / rocBLAS) on the same GEMM shapes as
tools/gemm_bench.py, bf16, bias fused by the library where it can."""
import sys
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M_lb, M_vv = 4112 * B, 3137 * B
g = torch.Generator(device=dev).manual_seed(0)
for name, M, N, K in [("lb.qkv", M_lb, 3072, 1024), ("lb.proj", M_lb, 1024, 1024), ("lb.fc1", M_lb, 4096, 1024), ("lb.fc2", M_lb, 1024, 4096),
                      ("vv.qkv", M_vv, 2304, 768), ("vv.proj", M_vv, 768, 768), ("vv.fc1", M_vv, 3072, 768), ("vv.fc2", M_vv, 768, 3072)]:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K**-0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
    best = 1e9
    for _ in range(3):
        F.linear(a, w, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            F.linear(a, w, b)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print(f"{name:8s} M={M:6d} N={N:5d} K={K:5d}: {best*1e3:7.1f} us {2.0*M*N*K/best/1e9:7.1f} TF", flush=True)
