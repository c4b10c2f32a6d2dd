#!/usr/bin/env python3
"""MXFP8 vs bf16 GEMM on the encoder shapes (B=8): time of the GEMM alone and of quantise(A) + GEMM."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from merv_amd import ops

dev = torch.device("cuda:0")
B = 8
M_lb, M_vv = 4112 * B, 3137 * B
shapes = [("lb.qkv", M_lb, 3072, 1024), ("lb.proj", M_lb, 1024, 1024), ("lb.fc1", M_lb, 4096, 1024), ("lb.fc2", M_lb, 1024, 4096),
          ("vv.qkv", M_vv, 2304, 768), ("vv.proj", M_vv, 768, 768), ("vv.fc1", M_vv, 3072, 768), ("vv.fc2", M_vv, 768, 3072)]
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, n=10):
    fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for name, M, N, K in shapes:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K**-0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    aq, asc = ops.quantize_mxfp8(a)
    wq, wsc = ops.quantize_mxfp8(w)
    t_bf = timeit(lambda: ops.gemm(a, w, bias=bias, out=out))
    t_mx = timeit(lambda: ops.gemm_mxfp8(aq, asc, wq, wsc, bias=bias, out=out))
    t_q = timeit(lambda: ops.quantize_mxfp8(a))
    err = float((ops.gemm_mxfp8(aq, asc, wq, wsc).float() - a.float() @ w.float().t()).norm() / (a.float() @ w.float().t()).norm())
    fl = 2.0 * M * N * K
    print(f"{name:8s} M={M:6d} N={N:5d} K={K:5d} | bf16 {t_bf:7.1f} us {fl/t_bf/1e6:7.1f} TF | mxfp8 {t_mx:7.1f} us {fl/t_mx/1e6:7.1f} TF"
          f" | quantise A {t_q:6.1f} us | rel err vs exact {err:.4f}")
