#!/usr/bin/env python3
"""Is the HIP decode step as close to the exact function as PyTorch-ROCm's bf16 decode step? Llama-2-7B geometry, random
init: the same model in fp32 (the checker) and in bf16 through (a) merv_amd.llm.StaticDecoder (PyTorch-ROCm ops) and (b)
merv_amd.llm.HipDecoder (csrc/decode.hip); identical prefill, identical tokens; logits compared at the same positions."""
import copy
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from merv_amd.llm import HipDecoder, LlamaBackbone, StaticDecoder, llama2_7b_config

dev = torch.device("cuda:0")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = dict(llama2_7b_config(), num_hidden_layers=layers)
llm = LlamaBackbone(cfg, device=dev)
m32 = copy.deepcopy(llm.llm).float()
emb = (torch.randn(1, 1049, 4096, generator=torch.Generator().manual_seed(0)) * 0.02).to(torch.bfloat16).to(dev)
d32, dpt, dhip = StaticDecoder(m32, 1280, 1), StaticDecoder(llm.llm, 1280, 1), HipDecoder(llm.llm, 1280, 1)
l32, lpt, lhip = d32.prefill(emb.float()), dpt.prefill(emb), dhip.prefill(emb)
rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
res = {"layers": layers, "prefill_pytorch_bf16_vs_fp32": round(rel(lpt, l32), 5), "steps": []}
for _ in range(6):
    tok = l32.argmax(-1)
    l32, lpt, lhip = d32.decode(tok, use_graph=False).clone(), dpt.decode(tok, use_graph=False).clone(), dhip.decode(tok, use_graph=False).clone()
    res["steps"].append({"pytorch_bf16_vs_fp32": round(rel(lpt, l32), 5), "hip_vs_fp32": round(rel(lhip, l32), 5), "hip_vs_pytorch_bf16": round(rel(lhip, lpt), 5),
                         "argmax": [int(l32.argmax()), int(lpt.argmax()), int(lhip.argmax())]})
print(json.dumps(res))
