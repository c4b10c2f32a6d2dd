#!/usr/bin/env python3
"""Does a weight matrix that was just read come back faster (Infinity Cache)? GEMV of one [N, K] matrix out of a ring of R
copies, graph of 16 launches: R = 1 re-reads the same 33 / 90 MB, R = 4 stays inside 256 MB, R = 16 does not."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from merv_amd import _lib
from merv_amd._lib import check, ptr

dev = torch.device("cuda:0")
lib = _lib.load()
res = {}
for (n, k) in [(4096, 4096), (4096, 11008)]:
    for R in (1, 2, 4, 16):
        ws = [torch.randn(n, k, device=dev, dtype=torch.bfloat16) for _ in range(R)]
        x = torch.randn(k, device=dev, dtype=torch.bfloat16)
        y = torch.empty(n, device=dev, dtype=torch.bfloat16)
        side = torch.cuda.Stream(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            check(lib.merv_decode_gemv(ptr(ws[0]), 0, ptr(x), 0, ptr(y), 0, n, k, 0, 0.0, side.cuda_stream), "gemv")
            side.synchronize()
            with torch.cuda.graph(g, stream=side):
                for i in range(16):
                    check(lib.merv_decode_gemv(ptr(ws[i % R]), 0, ptr(x), 0, ptr(y), 0, n, k, 0, 0.0, torch.cuda.current_stream(dev).cuda_stream), "gemv")
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 20 / 16
        res[f"{n}x{k} ring {R} ({R * n * k * 2 / 1e6:.0f} MB)"] = {"us": round(t * 1e6, 2), "TB_per_s": round(n * k * 2 / t / 1e12, 2)}
        del ws
print(json.dumps(res, indent=1))
