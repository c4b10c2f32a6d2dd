#!/usr/bin/env python3
"""Per-launch time of each kernel class of the HIP decode step (Llama-2-7B geometry, position 1049): a graph of the 32 layers'
launches of ONE class (the layers' own weights, so nothing is cache-resident), replayed; the sum against the whole step."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from merv_amd._lib import check, ptr
from merv_amd.llm import HipDecoder, LlamaBackbone, llama2_7b_config

def main():
    dev = torch.device("cuda:0")
    llm = LlamaBackbone(llama2_7b_config(), device=dev)
    import os
    HipDecoder.NSPLIT = int(os.environ.get("DEC_NSPLIT", HipDecoder.NSPLIT))
    d = HipDecoder(llm.llm, 1280, 1)
    emb = torch.randn(1, 1049, 4096, device=dev, dtype=torch.bfloat16) * 0.02
    tok = d.prefill(emb).argmax(-1)
    d.decode(tok)
    lib, m = d.lib, d.m
    D, I, H, Hkv, hd = d.cfg.hidden_size, d.cfg.intermediate_size, d.H, d.Hkv, d.hd
    x, pos = ptr(d.x), ptr(d.pos)
    side = torch.cuda.Stream(dev)


    def timeit(fn, n=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n


    def graph_of(per_layer):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            st = side.cuda_stream
            for lyr in m.model.layers[:2]:
                per_layer(lyr, 0, st)
            side.synchronize()
            with torch.cuda.graph(g, stream=side):
                st = torch.cuda.current_stream(dev).cuda_stream
                for li, lyr in enumerate(m.model.layers):
                    per_layer(lyr, li, st)
        return g


    def gemv(W, W2, xin, res, y, N, K, st, norm=None):
        check(lib.merv_decode_gemv(ptr(W), 0 if W2 is None else ptr(W2), xin, res, y, 0, N, K, 0 if norm is None else ptr(norm), d.eps, st), "gemv")


    scratch = torch.empty(D, dtype=torch.bfloat16, device=dev)
    classes = {
        "qkv (3 GEMV, norm fused)": lambda l, li, st: check(lib.merv_decode_gemv3(
            ptr(l.self_attn.q_proj.weight), ptr(l.self_attn.k_proj.weight), ptr(l.self_attn.v_proj.weight), x, ptr(d.q), ptr(d.k), ptr(d.v),
            H * hd, Hkv * hd, Hkv * hd, D, ptr(l.input_layernorm.weight), d.eps, st), "g3"),
        "rotary + cache + attention + merge": lambda l, li, st: check(lib.merv_decode_attention_fused(
            ptr(d.q), ptr(d.k), ptr(d.v), ptr(d.cos), ptr(d.sin), pos, ptr(d.K[li]), ptr(d.V[li]), ptr(d.ao), ptr(d.ws), H, Hkv, hd, d.max_len,
            d.NSPLIT, hd**-0.5, st), "attn"),
        "o_proj + residual": lambda l, li, st: gemv(l.self_attn.o_proj.weight, None, ptr(d.ao), x, ptr(scratch), D, H * hd, st),
        "rotary + cache + split attention (no merge)": lambda l, li, st: check(lib.merv_decode_attention_split(
            ptr(d.q), ptr(d.k), ptr(d.v), ptr(d.cos), ptr(d.sin), pos, ptr(d.K[li]), ptr(d.V[li]), ptr(d.ws), H, Hkv, hd, d.max_len,
            d.NSPLIT, hd**-0.5, st), "split"),
        "o_proj merging the splits + residual": lambda l, li, st: check(lib.merv_decode_oproj_merge(
            ptr(l.self_attn.o_proj.weight), x, ptr(scratch), ptr(d.ws), 0, D, H, hd, d.NSPLIT, st), "om"),
        "gate / up (norm, silu*up fused)": lambda l, li, st: gemv(l.mlp.gate_proj.weight, l.mlp.up_proj.weight, x, 0, ptr(d.mid), I, D, st,
                                                                   norm=l.post_attention_layernorm.weight),
        "down_proj + residual": lambda l, li, st: gemv(l.mlp.down_proj.weight, None, ptr(d.mid), x, ptr(scratch), D, I, st),
    }
    res = {}
    total = 0.0
    for name, fn in classes.items():
        g = graph_of(fn)
        t = timeit(g.replay) / len(m.model.layers)
        res[name] = round(t * 1e6, 2)
        if "one launch" not in name and "split" not in name:
            total += t
    res["sum_per_layer_us"] = round(total * 1e6, 2)
    res["step_graph_ms"] = round(timeit(lambda: d.decode(tok)) * 1e3, 3)
    res["step_minus_32_layers_ms"] = round(res["step_graph_ms"] - total * 32 * 1e3, 3)
    res["nsplit"] = HipDecoder.NSPLIT
    print(json.dumps(res))


with torch.inference_mode():
    main()
