#!/usr/bin/env python3
"""Deferred-max threshold of the attention kernels (AttnArgs::rescale_thr): MERV_ATTN_RESCALE_THR = 0 (exact running maximum,
rounds 1-2) against the shipped 8 and a few more, interleaved rounds in one process (the launcher re-reads the variable per
launch), encoder shapes on random data; also the difference between the outputs (two roundings of the same function)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import os
os.environ.setdefault("MERV_TUNING_HOOKS", "1")  # forced tile configurations / kernel forms: the hooks build (merv_amd/_lib.py)
import torch

from merv_amd import _lib, ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
thrs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "4", "8", "16"]
g = torch.Generator(device=dev).manual_seed(0)
for name, nseq, L, heads in [("languagebind", 16 * B, 257, 16), ("dinov2", 16 * B, 261, 16), ("siglip", 16 * B, 196, 12), ("vivit", B, 3137, 12)]:
    D = heads * 64
    qkv = (torch.randn(nseq * L, 3 * D, generator=g, device=dev) * 1.5).to(torch.bfloat16)
    times, outs = {t: [] for t in thrs}, {}
    for rnd in range(4):
        for t in thrs:
            _lib.load().merv_debug_set_attn_rescale_thr(float(t))
            o = ops.attention(qkv, nseq, L, heads)
            if rnd == 0:
                outs[t] = o.float().clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.attention(qkv, nseq, L, heads)
            e1.record()
            torch.cuda.synchronize()
            times[t].append(e0.elapsed_time(e1) / 10)
    _lib.load().merv_debug_set_attn_rescale_thr(8.0)
    base = outs[thrs[0]]
    print(f"attn {name:13s} L={L:5d}: " + " | ".join(f"thr {t}: {min(v)*1e3:7.1f} us  d={float((outs[t]-base).norm()/base.norm()):.1e}" for t, v in times.items()), flush=True)
