#!/usr/bin/env python3
"""Small-batch regime: how long is each encoder's chain alone, in pairs, and all four together, at B videos per step?
`batch_chains.py B [B ...]`: pipelined rate (enqueue n steps, one sync) of (a) the product step, (b) every encoder's a4-a9 chain alone on
its stream, (c) LanguageBind + DINOv2 only, (d) the three smaller encoders only. The step can be no shorter than its longest chain alone;
the gap between the two is what the chains cost each other."""
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
names = [s.name for s in specs]


def rate(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3)


def subset(idx, pix):
    """the chains of encoders `idx` as the step runs them (own streams, event-joined), nothing else"""
    main = torch.cuda.current_stream(dev)
    B = pix[0].shape[0]
    smap = path.stream_map(B)
    start = torch.cuda.Event(); start.record(main)
    for i in path.enqueue_order(B):
        if i not in idx:
            continue
        st = path.streams[smap[i]]
        st.wait_event(start)
        path.encode_project(i, pix[i], st)
        ev = torch.cuda.Event(); ev.record(st)
        main.wait_event(ev)


out = {}
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    pix = bench.synth_pixels(specs, B, dev, seed=0)
    r = {"step_ms": rate(lambda: path.forward(pix))}
    for i, n in enumerate(names):
        r[f"{n}_alone_ms"] = rate(lambda: subset({i}, pix))
    r["languagebind+dinov2_ms"] = rate(lambda: subset({0, 1}, pix))
    r["dinov2+vivit+siglip_ms"] = rate(lambda: subset({1, 2, 3}, pix))
    r["stream_map"] = path.stream_map(B)
    path.concurrent = False
    r["one_stream_ms"] = rate(lambda: path.forward(pix))
    path.concurrent = True
    out[f"B={B}"] = r
    print(f"B={B}", json.dumps(r), flush=True)
print(json.dumps(out))
