#!/usr/bin/env python3
"""VERDICT r5 item 7, one bounded experiment: every GEMM's remaining-rows launch forked onto a sibling stream that owns an otherwise idle hardware
queue (event-forked before the main launch, event-joined behind it), so that it overlaps its own main launch's tail instead of following it.
Hooks build (merv_debug_set_rest_fork is a no-op in the product). At 16 videos the stream map is 0111 (LanguageBind alone, the other three back to
back): side streams 2 and 3 are idle and become the siblings; at 4 videos (map 0112) only stream 3 is, and goes to LanguageBind's chain.
Three alternating pairs per batch size; bits must not change."""
import json
import os
import sys
import time
from pathlib import Path

os.environ["MERV_TUNING_HOOKS"] = "1"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

import bench
from merv_amd import _lib

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
lib = _lib.load()
assert lib.merv_tuning_hooks() == 1
specs, _, path, _ = bench.build_models(dev)


def rate(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3), out


res = {}
for B in (16, 4):
    pix = bench.synth_pixels(specs, B, dev, seed=0)
    smap = path.stream_map(B)
    used = sorted(set(smap))
    idle = [i for i in range(len(path.streams)) if i not in used]
    pairs = []
    ref = None
    for rep in range(3):
        lib.merv_debug_set_rest_fork(0, 0)
        t_off, out = rate(lambda: path.forward(pix))
        fused_off = out[0].clone()
        for k, st in enumerate(used[: len(idle)]):  # the chains in cost order get the idle queues
            lib.merv_debug_set_rest_fork(path.streams[st].cuda_stream, path.streams[idle[k]].cuda_stream)
        t_on, out = rate(lambda: path.forward(pix))
        fused_on = out[0].clone()
        lib.merv_debug_set_rest_fork(0, 0)
        pairs.append({"plain_ms": t_off, "forked_ms": t_on, "same_bits": bool(torch.equal(fused_on, fused_off))})
    res[f"{B} videos"] = {"stream_map": smap, "sibling_streams": {str(used[k]): idle[k] for k in range(min(len(used), len(idle)))}, "pairs": pairs}
    print(B, json.dumps(res[f"{B} videos"]), flush=True)
print(json.dumps(res))
