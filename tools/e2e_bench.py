#!/usr/bin/env python3
"""Secondary metric of BASELINE.json ("e2e gen tok/s, merv-full 16-frame"): quick_start-shaped flow on one MI355X with
random-init weights of the named architectures (no checkpoints here): decoded uint8 clip in HBM -> GPU transforms ->
4 HIP encoders -> projectors -> fusion -> splice -> Llama-2-7B (PyTorch-ROCm, SDPA) prefill of 1024 + prompt tokens ->
greedy decode of N tokens. Prints one JSON line with the stage times."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from merv_amd.backbones import VIDEO_BACKBONES
from merv_amd.llm import LlamaBackbone, llama2_7b_config
from merv_amd.vidlm import MERV

dev = torch.device("cuda:0")
new_tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ids = ["languagebind-video-noclass", "dinov2-video-all-tokens", "vivit-google-b-all-no-cls-16frames", "siglip-vit-b16-224px-all-no-cls"]
frames = [16, 16, 32, 16]
bbs = [VIDEO_BACKBONES[i]["cls"](i, "resize-naive", num_frames=f, weights="random", device=dev, **VIDEO_BACKBONES[i]["kwargs"])
       for i, f in zip(ids, frames)]
llm = LlamaBackbone(llama2_7b_config(), device=dev)
llm.config.eos_token_id = None  # random weights: never stop early
m = MERV(bbs, llm)
clip = (torch.randint(0, 256, (300, 360, 640, 3), dtype=torch.uint8), 29.97)
prompt = [1] + list(range(100, 124))  # BOS + 24 prompt tokens


def run():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.generate(clip, prompt, frames, max_new_tokens=new_tokens)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


run()
t, out = run()
# stage split: visual branch alone, from decoded frames already on the device
from merv_amd.sampler import temporal_subsample
from merv_amd.video_io import load_video
fr = load_video(clip, num_frames=32).to(dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
vv = [vb.video_transform(fr[temporal_subsample(32, 32, nf)].contiguous())[None] for vb, nf in zip(bbs, frames)]
torch.cuda.synchronize(); t_pre = time.perf_counter() - t0
t0 = time.perf_counter(); fused, w = m.encode(vv); torch.cuda.synchronize(); t_enc = time.perf_counter() - t0
print(json.dumps({"what": "e2e quick_start-shaped generate(), merv-full geometry, random-init Llama-2-7B geometry bf16 (PyTorch-ROCm SDPA)",
                  "new_tokens": int(out.shape[1]), "total_s": round(t, 3), "e2e_generated_tok_per_s": round(out.shape[1] / t, 2),
                  "gpu_transforms_ms": round(t_pre * 1e3, 2), "encoders_projectors_fusion_ms": round(t_enc * 1e3, 2),
                  "prefill_tokens": 1024 + len(prompt)}))
