#!/usr/bin/env python3
"""Does a weight matrix that was just read come back faster (Infinity Cache)? GEMV of one [N, K] matrix out of a ring of R
copies, graph of 16 launches: R = 1 re-reads the same 33 / 90 MB, R = 4 stays inside 256 MB, R = 16 does not.
Round 6 (the bounding probe for weight prefetch under the decode step's latency-bound launches): the decode step's own
launch shapes (q / k / v as one 12288 x 4096 stream with the fused norm, o-projection 4096 x 4096, gate / up pair 2 x 11008 x 4096, down
4096 x 11008), each timed (a) cold -- the matrix comes from HBM: 600 MB of other traffic since it was last read -- and (b) touched by a
read-only pass on a SIDE stream immediately before (event-joined), i.e. resident in the 256 MB memory-side cache as far as a prefetch can
make it. If (b) is not >= 15 % faster than (a) for the o-projection and q / k / v, prefetching cannot pay."""
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

# ---- round 6: cold against touched-from-a-side-stream, the decode step's launch shapes, HIP events around the one launch
flush = torch.empty(600 * 1024 * 1024 // 2, dtype=torch.bfloat16, device=dev).normal_()
side2 = torch.cuda.Stream(dev)
main = torch.cuda.current_stream(dev)


def launch(kind, W, W2, x, y, n, k, norm):
    st = main.cuda_stream
    if kind == "qkv":
        check(lib.merv_decode_gemv3_bias(ptr(W[0]), ptr(W[1]), ptr(W[2]), ptr(x), ptr(y[0]), ptr(y[1]), ptr(y[2]), n, n, n, k, ptr(norm), 1e-5, 0, 0, 0, st), "gemv3")
    else:
        check(lib.merv_decode_gemv(ptr(W), 0 if W2 is None else ptr(W2), ptr(x), 0, ptr(y), 0, n, k, 0 if norm is None else ptr(norm), 1e-5, st), "gemv")


bound = {}
for name, kind, n, k in [("o_proj 4096x4096", "plain", 4096, 4096), ("q/k/v 3 x 4096x4096 + norm", "qkv", 4096, 4096),
                         ("gate/up 2 x 11008x4096 + norm", "pair", 11008, 4096), ("down 4096x11008", "plain", 4096, 11008)]:
    x = torch.randn(k, device=dev, dtype=torch.bfloat16)
    norm = torch.ones(k, device=dev, dtype=torch.bfloat16)
    if kind == "qkv":
        W = [torch.randn(n, k, device=dev, dtype=torch.bfloat16) for _ in range(3)]; W2 = None
        y = [torch.empty(n, device=dev, dtype=torch.bfloat16) for _ in range(3)]; touch = W; nbytes = 3 * n * k * 2; nrm = norm
    elif kind == "pair":
        W = torch.randn(n, k, device=dev, dtype=torch.bfloat16); W2 = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
        y = torch.empty(n, device=dev, dtype=torch.bfloat16); touch = [W, W2]; nbytes = 2 * n * k * 2; nrm = norm
    else:
        W = torch.randn(n, k, device=dev, dtype=torch.bfloat16); W2 = None
        y = torch.empty(n, device=dev, dtype=torch.bfloat16); touch = [W]; nbytes = n * k * 2; nrm = None
    ts = {"cold": [], "touched": []}
    for rep in range(12):
        for mode in ("cold", "touched"):
            flush.add_(1.0)  # 1.2 GB of read + write traffic: nothing of W is left in L2 or the memory-side cache
            if mode == "touched":
                done = torch.cuda.Event()
                side2.wait_stream(main)
                with torch.cuda.stream(side2):
                    for t in touch:
                        t.view(torch.int16).max()  # read-only pass over the matrix
                    done.record(side2)
                main.wait_event(done)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main); launch(kind, W, W2, x, y, n, k, nrm); e1.record(main)
            torch.cuda.synchronize()
            if rep >= 2:
                ts[mode].append(e0.elapsed_time(e1) * 1e3)
    c, t = sorted(ts["cold"])[len(ts["cold"]) // 2], sorted(ts["touched"])[len(ts["touched"]) // 2]
    bound[name] = {"MB": round(nbytes / 1e6, 1), "cold_us": round(c, 2), "touched_us": round(t, 2), "cold_TB_s": round(nbytes / c / 1e6, 2),
                   "touched_TB_s": round(nbytes / t / 1e6, 2), "speedup": round(c / t, 3)}
    del W, W2, touch
res["round6_cold_vs_touched_from_side_stream (HIP events around one launch: includes ~2 us of event / launch boundary)"] = bound
print(json.dumps(res, indent=1))
