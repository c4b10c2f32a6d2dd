#!/usr/bin/env python3
"""Small batches: does a high-priority stream for the largest encoder's chain (LanguageBind: the chain that ends the step) shorten the step?
`bench.py`-shaped pipelined rate at 1 / 2 / 4 videos with the path's side stream 0 (cost rank 0) replaced by a high-priority stream, alternating."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
specs, _, path, _ = bench.build_models(dev)
plain = list(path.streams)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
hp = torch.cuda.Stream(dev, priority=-1)
with torch.cuda.stream(hp):  # bind it to its hardware queue now (HIP binds at first use), from this thread
    torch.zeros(1, device=dev)
hp.synchronize()


def rate(fn, n=30, warm=6):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3)


res = {}
for B in (1, 2, 4):
    pix = bench.synth_pixels(specs, B, dev, seed=0)
    rows = []
    for rep in range(3):
        path.streams = list(plain)
        a = rate(lambda: path.forward(pix))
        path.streams = [hp] + plain[1:]
        b = rate(lambda: path.forward(pix))
        rows.append({"plain_ms": a, "largest_chain_high_priority_ms": b})
    path.streams = list(plain)
    res[f"{B} videos"] = rows
    print(B, rows, flush=True)
print(json.dumps(res))
