#!/usr/bin/env python3
"""Decode step with the o-projection's weights touched by extra workgroups of the split-attention launch (each on the XCD whose L2 the
o-projection's workgroup of the same index reads through) against the plain launch: graph-replayed greedy steps at positions 1049.., Llama-2-7B
geometry, three alternating pairs; the tokens must not change."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from merv_amd.llm import HipDecoder, LlamaBackbone, llama2_7b_config

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
llm = LlamaBackbone(llama2_7b_config(), device=dev)
llm.config.eos_token_id = None
n_pre = 1049
emb = (torch.randn(1, n_pre, llm.config.hidden_size, device=dev, dtype=torch.bfloat16) * 0.02)
rows = []
toks = {}
with torch.inference_mode():
    dec = HipDecoder(llm.llm, 1280, 1)
    for rep in range(3):
        row = {}
        for mode in (False, True):
            dec.prefetch_oproj = mode
            dec.greedy_graph = dec.greedy_graph_chunk = None
            lg = dec.prefill(emb)
            tok = lg.argmax(-1)
            out = dec.greedy_run(tok, 16, n_pre)
            toks.setdefault(mode, out.clone())
            assert torch.equal(out, toks[mode])
            dec.prefill(emb)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            dec.greedy_run(tok, 64, n_pre)
            torch.cuda.synchronize()
            row["prefetch_wo" if mode else "plain"] = round((time.perf_counter() - t0) / 64 * 1e3, 4)
        rows.append(row)
        print(row, flush=True)
res = {"ms_per_decoded_token": rows, "same_tokens": bool(torch.equal(toks[False], toks[True]))}
print(json.dumps(res))
