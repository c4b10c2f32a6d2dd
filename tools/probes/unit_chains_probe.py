#!/usr/bin/env python3
"""Bounding probe for frame-range unit chains at small batch: does running LanguageBind as two clip chains and DINOv2 as
two frame-range chains (six or five concurrent chains instead of four) shorten the step? Two path objects over the same seeded weights give
each half its own workspace; outputs go to the paths' own buffers (placement of the rows is not part of the timing).
Run with GPU_MAX_HW_QUEUES=8 (HIP folds streams onto 4 hardware queues by default) and without."""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
specs, _, p1, _ = bench.build_models(dev)
_, _, p2, _ = bench.build_models(dev)
streams = [torch.cuda.Stream(dev) for _ in range(8)]  # created AFTER the paths' own four


def rate(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 3)


def halves(pix, spec):
    h = spec.frames // 2
    if spec.pix_layout == "BCFHW":
        return pix[:, :, :h].contiguous(), pix[:, :, h:].contiguous()
    return pix[:, :h].contiguous(), pix[:, h:].contiguous()


def run(chains):
    """chains: list of (path, encoder index, pixels, frames | None, stream)"""
    main = torch.cuda.current_stream(dev)
    start = torch.cuda.Event(); start.record(main)
    seen = set()
    for path, i, pix, nf, st in chains:
        if st not in seen:
            st.wait_event(start); seen.add(st)
        path.encode_project(i, pix, st, frames=nf)
    for st in seen:
        ev = torch.cuda.Event(); ev.record(st); main.wait_event(ev)


out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    pix = bench.synth_pixels(specs, B, dev, seed=0)
    lb0, lb1 = halves(pix[0], specs[0])
    dn0, dn1 = halves(pix[1], specs[1])
    S = p1.streams + streams  # the product's four side streams first
    r = {"product_step_ms": rate(lambda: p1.forward(pix))}
    four = [(p1, 0, pix[0], None, S[0]), (p1, 1, pix[1], None, S[1]), (p1, 2, pix[2], None, S[2]), (p1, 3, pix[3], None, S[3])]
    r["four_chains_own_streams_ms"] = rate(lambda: run(four))
    six = [(p1, 0, lb0, 8, S[0]), (p2, 0, lb1, 8, S[1]), (p1, 1, dn0, 8, S[2]), (p2, 1, dn1, 8, S[3]), (p1, 2, pix[2], None, S[4]), (p1, 3, pix[3], None, S[5])]
    r["six_chains_lb2_dino2_ms"] = rate(lambda: run(six))
    five = six[:4] + [(p1, 2, pix[2], None, S[4]), (p1, 3, pix[3], None, S[4])]
    r["five_chains_vivit_siglip_shared_ms"] = rate(lambda: run(five))
    five_b = [(p1, 0, lb0, 8, S[0]), (p2, 0, lb1, 8, S[1]), (p1, 1, pix[1], None, S[2]), (p1, 2, pix[2], None, S[3]), (p1, 3, pix[3], None, S[4])]
    r["five_chains_lb2_only_ms"] = rate(lambda: run(five_b))
    lb_only = [(p1, 0, lb0, 8, S[0]), (p2, 0, lb1, 8, S[1])]
    r["languagebind_two_clip_chains_alone_ms"] = rate(lambda: run(lb_only))
    r["languagebind_one_chain_alone_ms"] = rate(lambda: run(four[:1]))
    dn_only = [(p1, 1, dn0, 8, S[2]), (p2, 1, dn1, 8, S[3])]
    r["dinov2_two_chains_alone_ms"] = rate(lambda: run(dn_only))
    r["dinov2_one_chain_alone_ms"] = rate(lambda: run(four[1:2]))
    out[f"B={B}"] = r
    print(f"B={B}", json.dumps(r), flush=True)
print(json.dumps(out))
