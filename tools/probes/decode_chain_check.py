#!/usr/bin/env python3
"""The one-launch decode step and the fused attention + o-projection launch against the 5-launches-per-layer sequence: Llama-2-7B geometry (random init), a 1049-token prefill, then
N greedy steps through both; logits must be bit-identical; time per step of each (graph-replayed)."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from merv_amd.llm import HipDecoder, LlamaBackbone, llama2_7b_config


def main():
    dev = torch.device("cuda:0")
    cfg = llama2_7b_config()
    if len(sys.argv) > 1:
        cfg["num_hidden_layers"] = int(sys.argv[1])
    llm = LlamaBackbone(cfg, device=dev)
    emb = torch.randn(1, 1049, cfg["hidden_size"], device=dev, dtype=torch.bfloat16) * 0.02
    out = {}
    logits = {}
    for name, chain, fuse in (("launches", False, False), ("attn_oproj", False, True), ("chain", True, False)):
        HipDecoder.use_chain, HipDecoder.use_attn_oproj = chain, fuse
        d = HipDecoder(llm.llm, 1280, 1)
        tok = d.prefill(emb).argmax(-1)
        ls = []
        for _ in range(6):
            lg = d.decode(tok)
            ls.append(lg.clone())
            tok = lg.argmax(-1)
        torch.cuda.synchronize()
        logits[name] = torch.stack(ls)
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            d.decode(tok)
        torch.cuda.synchronize()
        out[name + "_ms_per_step"] = round((time.perf_counter() - t0) / n * 1e3, 4)
        out[name + "_err"] = int(d.chain_err.item())
        del d
    for name in ("attn_oproj", "chain"):
        out[name + "_bit_identical"] = bool(torch.equal(logits["launches"], logits[name]))
        out[name + "_max_abs_diff"] = float((logits["launches"] - logits[name]).abs().max())
    out["finite"] = bool(torch.isfinite(logits["chain"]).all() and torch.isfinite(logits["attn_oproj"]).all())
    print(json.dumps(out))


with torch.inference_mode():
    main()
