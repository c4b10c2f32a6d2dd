#!/usr/bin/env python3
"""Single-call latency of the visual path at batch 1 (what one generate() pays), eager vs hipGraph replay, and the
back-to-back rate (what bench.py --batch 1 reports: host launches of call i+1 overlap the GPU work of call i)."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
specs, bbs, path, extras = bench.build_models(dev)
if "--threads" in sys.argv:  # MervVisualPath.threaded_enqueue: 1 = one host thread per chain, 0 = one thread (default: the path's own rule)
    path.threaded_enqueue = sys.argv[sys.argv.index("--threads") + 1] == "1"
pix = bench.synth_pixels(specs, 1, dev, seed=0)


def single(fn, n=20):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


def pipelined(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    path.forward(pix)
res = {"eager_single_ms_median_min": single(lambda: path.forward(pix)), "eager_pipelined_ms": pipelined(lambda: path.forward(pix))}
t0 = time.perf_counter()
for _ in range(5):
    path.forward(pix)
res["eager_host_launch_ms"] = (time.perf_counter() - t0) / 5 * 1e3  # host time to enqueue one step (GPU lags behind)
torch.cuda.synchronize()
replay = path.capture(pix)
res["graph_single_ms_median_min"] = single(replay)
res["graph_pipelined_ms"] = pipelined(replay)
path.concurrent = False
res["eager_sequential_single_ms_median_min"] = single(lambda: path.forward(pix))
print(json.dumps(res))
