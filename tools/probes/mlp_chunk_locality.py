#!/usr/bin/env python3
"""Does producer -> consumer locality in the Infinity Cache pay for the MLP pair? The energy-bound probe (profiles/r05_gemm_energy_bound.json)
says the K = 4096 GEMM (fc2) runs 9-13 % faster when its A operand comes from L2 -- but A (the 539 MB fc1 output of LanguageBind at 16
videos) is compulsory traffic, read once from HBM. What CAN change is where it is read from: run fc1 -> fc2 over row chunks small
enough for the chunk's hidden activations to still sit in the 256 MB Infinity Cache when fc2 reads them. LanguageBind's MLP at 16 videos
(M = 65536 rows of the eight-phase part, D = 1024, hidden 4096, quick-GELU, residual), chunk sizes that keep every launch at complete
rounds of the chip (16384 rows = 1 round of fc2's 256 x 256 tiles, 4 of fc1's); modes interleaved in one process, medians."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from merv_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
M, D, H = 65536, 1024, 4096
x = torch.randn(M, D, generator=g, device=dev).to(torch.bfloat16)
w1 = (torch.randn(H, D, generator=g, device=dev) * D**-0.5).to(torch.bfloat16)
w2 = (torch.randn(D, H, generator=g, device=dev) * H**-0.5).to(torch.bfloat16)
b1, b2 = torch.randn(H, generator=g, device=dev) * 0.1, torch.randn(D, generator=g, device=dev) * 0.1
h = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
y = torch.empty(M, D, dtype=torch.bfloat16, device=dev)


def mlp(chunk):
    for r0 in range(0, M, chunk):
        ops.gemm(x[r0:r0 + chunk], w1, bias=b1, act="quick_gelu", out=h[r0:r0 + chunk])
        ops.gemm(h[r0:r0 + chunk], w2, bias=b2, res=x[r0:r0 + chunk], out=y[r0:r0 + chunk])


def only(which, chunk):
    for r0 in range(0, M, chunk):
        if which == 1:
            ops.gemm(x[r0:r0 + chunk], w1, bias=b1, act="quick_gelu", out=h[r0:r0 + chunk])
        else:
            ops.gemm(h[r0:r0 + chunk], w2, bias=b2, res=x[r0:r0 + chunk], out=y[r0:r0 + chunk])


def timeit(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


chunks = [65536, 32768, 16384, 8192]
for _ in range(20):
    mlp(M)  # warm the chip
res = {c: [] for c in chunks}
res1 = {c: [] for c in chunks}
res2 = {c: [] for c in chunks}
for rnd in range(7):
    for c in chunks[rnd % 4:] + chunks[:rnd % 4]:
        res[c].append(timeit(lambda: mlp(c)))
        res1[c].append(timeit(lambda: only(1, c)))
        res2[c].append(timeit(lambda: only(2, c)))
med = lambda v: sorted(v)[len(v) // 2]
out = {"what": "LanguageBind MLP pair at 16 videos (M = 65536), fc1 -> fc2 run over row chunks: us for the whole M", "hidden_bytes_per_chunk_MB": {c: c * H * 2 / 1e6 for c in chunks},
       "pair_us": {c: round(med(res[c]), 1) for c in chunks}, "fc1_alone_us": {c: round(med(res1[c]), 1) for c in chunks},
       "fc2_alone_us": {c: round(med(res2[c]), 1) for c in chunks}}
print(json.dumps(out))
