#!/usr/bin/env python3
"""Where a batch-1 Llama-2-7B decode step on PyTorch-ROCm spends its time: library GEMV rates per projection shape, the
graph-replayed step as a whole, and the step with the linear layers removed (everything that is not a weight stream)."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import torch.nn.functional as F

from merv_amd.llm import LlamaBackbone, StaticDecoder, llama2_7b_config

dev = torch.device("cuda:0")
res = {}


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


x = torch.randn(1, 1, 4096, device=dev, dtype=torch.bfloat16)
x2 = torch.randn(1, 1, 11008, device=dev, dtype=torch.bfloat16)
for name, (n, k, inp) in {"q/k/v/o 4096x4096": (4096, 4096, x), "gate/up 11008x4096": (11008, 4096, x),
                          "down 4096x11008": (4096, 11008, x2), "lm_head 32064x4096": (32064, 4096, x)}.items():
    ws = [torch.randn(n, k, device=dev, dtype=torch.bfloat16) for _ in range(8)]  # rotate: defeat the caches
    i = [0]

    def f():
        i[0] = (i[0] + 1) % 8
        return F.linear(inp, ws[i[0]])
    g = torch.cuda.CUDAGraph()
    f(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(8):
            f()
    t = timeit(g.replay, 20) / 8
    res[name] = {"us": round(t * 1e6, 1), "TB_per_s": round(n * k * 2 / t / 1e12, 2)}
    del ws

llm = LlamaBackbone(llama2_7b_config(), device=dev)
dec = StaticDecoder(llm.llm, 1280, 1)
emb = torch.randn(1, 1049, 4096, device=dev, dtype=torch.bfloat16) * 0.02
t0 = time.perf_counter(); logits = dec.prefill(emb); torch.cuda.synchronize(); res["prefill_1049_ms_first"] = round((time.perf_counter() - t0) * 1e3, 1)
t0 = time.perf_counter(); logits = dec.prefill(emb); torch.cuda.synchronize(); res["prefill_1049_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
tok = logits.argmax(-1)
dec.decode(tok)
res["decode_step_graph_ms"] = round(timeit(lambda: dec.decode(tok), 30) * 1e3, 3)
# the same step without the weight streams: F.linear replaced by a broadcast of zeros of the right shape
real_linear = F.linear


def fake_linear(inp, w, b=None):
    return inp[..., :1].expand(*inp.shape[:-1], w.shape[0]) * 0


dec2 = StaticDecoder(llm.llm, 1280, 1)
dec2.prefill(emb)
dec2.F = type("Fk", (), {"linear": staticmethod(fake_linear), "scaled_dot_product_attention": staticmethod(F.scaled_dot_product_attention),
                         "silu": staticmethod(F.silu)})
dec2.decode(tok)
res["decode_step_without_linears_graph_ms"] = round(timeit(lambda: dec2.decode(tok), 30) * 1e3, 3)
from merv_amd.llm import HipDecoder
hd = HipDecoder(llm.llm, 1280, 1)
hd.prefill(emb)
hd.decode(tok)
res["hip_decode_step_graph_ms"] = round(timeit(lambda: hd.decode(tok), 30) * 1e3, 3)
hd2 = HipDecoder(llm.llm, 1280, 1)
hd2.prefill(emb)
res["hip_decode_step_eager_ms"] = round(timeit(lambda: hd2.decode(tok, use_graph=False), 10) * 1e3, 3)
# logits of the two decoders at the SAME position after the same prefill (Llama-2-7B geometry, random init, 32 layers)
da, db = StaticDecoder(llm.llm, 1280, 1), HipDecoder(llm.llm, 1280, 1)
la, lb = da.prefill(emb), db.prefill(emb)
errs = []
for _ in range(4):
    t_ = la.argmax(-1)
    la, lb = da.decode(t_, use_graph=False).clone(), db.decode(t_, use_graph=False).clone()
    errs.append(float((lb - la).norm() / la.norm()))
res["hip_vs_pytorch_logits_rel_l2_4_steps"] = [round(e, 5) for e in errs]
del da, db
# the HIP GEMV per projection shape (weights rotated through 8 copies, graph of 8 calls)
from merv_amd import _lib
from merv_amd._lib import check, ptr
lib = _lib.load()
for name, (n, k, gated) in {"hip q/k/v/o 4096x4096": (4096, 4096, False), "hip gate+up fused 2x11008x4096": (11008, 4096, True),
                            "hip down 4096x11008": (4096, 11008, False), "hip lm_head 32064x4096": (32064, 4096, False)}.items():
    ws = [torch.randn(n, k, device=dev, dtype=torch.bfloat16) for _ in range(8)]
    xin = torch.randn(k, device=dev, dtype=torch.bfloat16)
    y = torch.empty(n, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream(dev).cuda_stream

    def call(i):
        check(lib.merv_decode_gemv(ptr(ws[i]), ptr(ws[(i + 1) % 8]) if gated else 0, ptr(xin), 0, ptr(y), 0, n, k, 0, 0.0,
                                   torch.cuda.current_stream(dev).cuda_stream), "gemv")
    call(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(8):
            call(i)
    t = timeit(g.replay, 20) / 8
    res[name] = {"us": round(t * 1e6, 1), "TB_per_s": round(n * k * 2 * (2 if gated else 1) / t / 1e12, 2)}
    del ws
res["weights_GB"] = round(sum(p.numel() for n, p in llm.llm.named_parameters() if "embed_tokens" not in n) * 2 / 1e9, 2)
print(json.dumps(res))
