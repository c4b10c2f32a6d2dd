#!/usr/bin/env python3
"""One video per call: the four chains enqueued by one host thread each (an encoder forward is ONE library call, made without the GIL) against the
one-thread enqueue. Single-call latency (sync, call, sync: what generate() pays; median / min of 30) and the back-to-back rate, alternating; the
fused tokens must keep their bits."""
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


def single(fn, n=30):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return round(ts[len(ts) // 2] * 1e3, 3), round(ts[0] * 1e3, 3)


def pipelined(fn, n=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3)


res = {}
for B in (1, 2):
    pix = bench.synth_pixels(specs, B, dev, seed=0)
    path.threaded_enqueue = False
    ref = path.forward(pix)[0].clone()
    rows = []
    for rep in range(3):
        row = {}
        for mode in (False, True):
            path.threaded_enqueue = mode
            for _ in range(3):
                out = path.forward(pix)[0]
            same = bool(torch.equal(out, ref))
            t0 = time.perf_counter()
            for _ in range(5):
                path.forward(pix)
            host = (time.perf_counter() - t0) / 5 * 1e3
            torch.cuda.synchronize()
            row["threads" if mode else "one_thread"] = {"single_call_ms_median_min": single(lambda: path.forward(pix)), "back_to_back_ms": pipelined(lambda: path.forward(pix)),
                                                        "host_enqueue_ms": round(host, 2), "same_bits": same}
        rows.append(row)
    res[f"{B} videos (stream map {path.stream_map(B)})"] = rows
    print(B, json.dumps(rows), flush=True)
path.threaded_enqueue = None
print(json.dumps(res))
