#!/usr/bin/env python3
"""What the MXFP8 mode costs and buys per choice of GEMMs (random-init weights of the merv-full geometry, full depth):
relative L2 of the fused visual tokens vs the bf16 path on the same inputs, and ms per 8-video step.
Writes gpurun_out/mx_accuracy.json."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
rel = lambda a, b: float((a - b).norm() / b.norm())


def timeit(path, pix, n=6):
    for _ in range(2):
        path.forward(pix)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        path.forward(pix)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


specs, path = bench.build_path(dev, concurrent=True)
pix = bench.synth_pixels(specs, 8, dev, seed=0)
ref_tok = [e.forward(p).float() for e, p in zip(path.encoders, pix)]
ref_fused = path.forward(pix)[0].float().clone()
rows = [{"gemms": "none (bf16)", "fused_rel_l2": 0.0, "ms_per_step": round(timeit(path, pix), 2)}]
for gemms in (("qkv", "proj", "fc1", "fc2"), ("fc1", "fc2"), ("qkv", "proj"), ("qkv", "fc1"), ("proj", "fc2"), ("fc2",)):
    for e in path.encoders:
        e.enable_mxfp8(gemms)
    fused = path.forward(pix)[0].float()
    row = {"gemms": "+".join(gemms), "fused_rel_l2": round(rel(fused, ref_fused), 4), "ms_per_step": round(timeit(path, pix), 2)}
    if len(gemms) == 4:
        row["encoder_tokens_rel_l2"] = {s.name: round(rel(e.forward(p).float(), r), 4) for s, e, p, r in zip(specs, path.encoders, pix, ref_tok)}
    rows.append(row)
    print(row, flush=True)
out = {"note": "MXFP8 mode vs bf16 path, same random-init weights and inputs, full depth (23/23/12/11 blocks), B=8; the GEMMs not "
               "listed stay bf16", "rows": rows}
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/mx_accuracy.json").write_text(json.dumps(out, indent=1))
