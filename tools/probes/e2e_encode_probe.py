#!/usr/bin/env python3
"""e2e_encode_probe.py: the visual path call of the e2e leg (bench.e2e_generate's inputs: a decoded uint8 clip through the backbones' own
frame transforms) against the same call on synthetic pixel tensors -- where do the extra milliseconds of `visual_path_ms` come from?"""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench
from merv_amd.vidlm import MERVVisual
from merv_amd.sampler import temporal_subsample
from merv_amd.video_io import load_video
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
specs, bbs, path, extras = bench.build_models(dev)
m = MERVVisual(bbs, llm_dim=4096)
clip = (torch.randint(0, 256, (300, 360, 640, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(5)), 29.97)
fr = load_video(clip, num_frames=max(bench.NUM_FRAMES)).to(dev)
vv = [vb.video_transform(fr[temporal_subsample(fr.shape[0], max(bench.NUM_FRAMES), nf)].contiguous())[None] for vb, nf in zip(bbs, bench.NUM_FRAMES)]
syn = [torch.randn(s.pixel_shape(1), device=dev).to(torch.bfloat16) for s in specs]
for name, x in (("transformed clip", vv), ("synthetic", syn), ("transformed clip", vv)):
    print(name, [(tuple(t.shape), str(t.dtype), t.is_contiguous()) for t in x])
    for _ in range(3): m.encode(x)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); m.encode(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("   min %.2f ms median %.2f ms" % (min(ts) * 1e3, sorted(ts)[5] * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    m.encode(vv); torch.cuda.synchronize()
rows = sorted(((e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0), key=lambda r: -r[2])
for k, c, t in rows[:14]: print("%9.1f us %5d  %s" % (t, c, k[:100]))
# the same call with the 7B LLM resident and after a generate() (what the bench's e2e leg times)
from merv_amd.llm import LlamaBackbone, llama2_7b_config
from merv_amd.vidlm import MERV
llm = LlamaBackbone(llama2_7b_config(), device=dev)
llm.config.eos_token_id = None
m2 = MERV(bbs, llm)
def lat(mm, x, label):
    for _ in range(3): mm.encode(x)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); mm.encode(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(label, "min %.2f ms median %.2f ms" % (min(ts) * 1e3, sorted(ts)[5] * 1e3))
lat(m, vv, "MERVVisual, LLM resident:")
lat(m2, vv, "MERV (with LLM) before generate:")
m2.generate(clip, [1] + list(range(100, 124)), bench.NUM_FRAMES, max_new_tokens=8)
lat(m2, vv, "MERV after generate:")
lat(m, vv, "MERVVisual after generate:")
m3 = MERVVisual(bbs, llm_dim=4096)
lat(m3, vv, "a second MERVVisual, built after the LLM:")
print("paths:", type(m._path).__name__, m._path.concurrent, m2._path.concurrent, m3._path.concurrent, getattr(m2, "concurrent", None))
import torch.nn as nn
print("MERV training?", m2.training, "MERVVisual training?", m.training, [p.requires_grad for p in m2.projectors[0].parameters()])
with torch.no_grad():
    lat(m2, vv, "MERV under no_grad:")
