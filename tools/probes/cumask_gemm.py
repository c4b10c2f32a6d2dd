#!/usr/bin/env python3
"""One encoder-block GEMM (LanguageBind qkv at 8 videos: M 16448, N 3072, K 1024) on CU-masked streams of different widths
(hipExtStreamCreateWithCUMask; bits [8u, 8u + 8) of the mask are one CU on every XCD, tools/probes/cumask.hip), alone and with
several disjoint partitions running at once, against plain streams. Outcome: DESIGN.md section 4."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from merv_amd import _lib, ops
from merv_amd._lib import check

dev = torch.device("cuda:0")
lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
M, N, K = 16448, 3072, 1024
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * K**-0.5).to(torch.bfloat16)
b = torch.zeros(N, device=dev)


def masked(units, start=0):
    mask = (C.c_uint32 * 8)()
    for bit in range(8 * start, 8 * (start + units)):
        mask[bit // 32] |= 1 << (bit % 32)
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=dev)


def bench(st, n=20):
    with torch.cuda.stream(st):
        out = ops.gemm(a, w, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(n):
            ops.gemm(a, w, b, out=out)
        e1.record(st)
    st.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("default stream        : %.1f us" % bench(torch.cuda.Stream(dev)))
for units in (32, 16, 14, 8, 4):
    print("masked, %2d units (%3d CUs): %.1f us" % (units, units * 8, bench(masked(units))))

# ---- concurrency: P disjoint partitions, each running the same GEMM n times, against the same total work on the whole chip
import time


def concurrent(parts, n=20):
    sts, start = [], 0
    for u in parts:
        sts.append(masked(u, start)); start += u
    outs = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in parts]
    def run():
        for st, o in zip(sts, outs):
            with torch.cuda.stream(st):
                for _ in range(n):
                    ops.gemm(a, w, b, out=o)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * len(parts)) * 1e6


def plain_streams(k, n=20):
    sts = [torch.cuda.Stream(dev) for _ in range(k)]
    outs = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(k)]
    def run():
        for st, o in zip(sts, outs):
            with torch.cuda.stream(st):
                for _ in range(n):
                    ops.gemm(a, w, b, out=o)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * k) * 1e6


print("per-GEMM time, 1 plain stream : %.1f us" % plain_streams(1))
print("per-GEMM time, 2 plain streams: %.1f us" % plain_streams(2))
print("per-GEMM time, 4 plain streams: %.1f us" % plain_streams(4))
for parts in ([16, 16], [8, 8, 8, 8], [4] * 8, [16, 8, 8], [14, 11, 4, 3]):
    print("per-GEMM time, partitions %s: %.1f us" % (parts, concurrent(parts)))
