#!/usr/bin/env python3
"""One video per step: the three smaller encoders' chains on CU-masked streams (hipExtStreamCreateWithCUMask: U of the 32 CU-octets, one CU per
XCD each), LanguageBind's chain unmasked -- does keeping part of the chip free for the chain that ends the step shorten it? Alternating with
the plain streams."""
import ctypes as C
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
hip = C.CDLL("libamdhip64.so")
specs, _, path, _ = bench.build_models(dev)
plain = list(path.streams)


def masked(units, start=0):
    mask = (C.c_uint32 * 8)()
    for bit in range(8 * start, 8 * (start + units)):
        mask[bit // 32] |= 1 << (bit % 32)
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=dev)


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
for B in (1, 2):
    pix = bench.synth_pixels(specs, B, dev, seed=0)
    smap = path.stream_map(B)
    lb_stream = smap[0]  # stream index of the largest encoder (index 0 = LanguageBind)
    rows = []
    for units in (32, 28, 24, 20, 16):
        ms = [masked(units, 32 - units) for _ in range(3)]  # the LAST `units` octets: the first 32 - units stay LanguageBind's alone
        row = {"cus_for_the_other_chains": units * 8}
        for rep in range(2):
            path.streams = list(plain)
            a = rate(lambda: path.forward(pix))
            k = 0
            st = list(plain)
            for i in range(len(st)):
                if i != lb_stream:
                    st[i] = ms[k]; k += 1
            path.streams = st
            b = rate(lambda: path.forward(pix))
            row[f"rep{rep}"] = {"plain_ms": a, "masked_ms": b}
        rows.append(row)
        print(B, row, flush=True)
    path.streams = list(plain)
    res[f"{B} videos"] = rows
print(json.dumps(res))
