#!/usr/bin/env python3
"""Reference point (measurement only): torch F.scaled_dot_product_attention on the encoder stack's attention shapes."""
import sys
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = torch.Generator(device=dev).manual_seed(0)
for name, nseq, L, heads in [("languagebind", 16 * B, 257, 16), ("dinov2", 16 * B, 261, 16), ("siglip", 16 * B, 196, 12), ("vivit", B, 3137, 12)]:
    q, k, v = [torch.randn(nseq, heads, L, 64, generator=g, device=dev).to(torch.bfloat16) for _ in range(3)]
    best = 1e9
    for _ in range(3):
        F.scaled_dot_product_attention(q, k, v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            F.scaled_dot_product_attention(q, k, v)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print(f"sdpa {name:13s} L={L:5d}: {best*1e3:8.1f} us  {4.0*nseq*L*L*heads*64/best/1e9:7.1f} TF", flush=True)
