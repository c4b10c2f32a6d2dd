#!/usr/bin/env python3
"""What the MXFP8 mode costs in accuracy at full depth (random-init weights of the merv-full geometry): relative L2 of
each encoder's output tokens and of the fused visual tokens, MXFP8 mode vs the default bf16 path on the same inputs.
Writes gpurun_out/mx_accuracy.json."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
specs, path = bench.build_path(dev, concurrent=True)
pix = bench.make_pixels(specs, 2, dev, seed=0) if hasattr(bench, "make_pixels") else None
if pix is None:
    g = torch.Generator().manual_seed(0)
    pix = [torch.randn(s.pixel_shape(2), generator=g).to(torch.bfloat16).to(dev) for s in specs]
ref_tok = [e.forward(p).float() for e, p in zip(path.encoders, pix)]
ref_fused = path.forward(pix)[0].float()
for e in path.encoders:
    e.enable_mxfp8()
mx_tok = [e.forward(p).float() for e, p in zip(path.encoders, pix)]
mx_fused = path.forward(pix)[0].float()
rel = lambda a, b: float((a - b).norm() / b.norm())
out = {"note": "MXFP8 mode vs bf16 path, same random-init weights and inputs, full depth (23/23/12/11 blocks), B=2",
       "encoder_tokens_rel_l2": {s.name: rel(a, b) for s, a, b in zip(specs, mx_tok, ref_tok)},
       "fused_tokens_rel_l2": rel(mx_fused, ref_fused)}
print(json.dumps(out))
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/mx_accuracy.json").write_text(json.dumps(out, indent=1))
