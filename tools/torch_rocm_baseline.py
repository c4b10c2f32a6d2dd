#!/usr/bin/env python3
"""What the reference's own stack does on this GPU: the same four encoders + projectors + fusion written with plain
PyTorch-ROCm ops in bf16 (F.linear -> hipBLASLt/rocBLAS, F.scaled_dot_product_attention, F.layer_norm, F.gelu), i.e. the
kernels timm / transformers would launch under `vidlm.to(bf16)` + autocast, encoders run one after the other
(merv.py:563-566). Measurement only -- nothing here is part of the product. Prints one JSON line."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.nn.functional as F

from merv_amd.encoder import merv_full_specs

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)


def rn(*shape, std=0.02):
    return (torch.randn(*shape, generator=g, device=dev) * std).to(dt)


def make(spec):
    D, Mh = spec.dim, spec.mlp_dim
    P = spec.s_out * (spec.t_out if spec.joint_space_time else 1)
    W = {"patch_w": rn(D, 3, *( [spec.tubelet] if spec.tubelet > 1 else []), spec.patch, spec.patch, std=0.03), "patch_b": rn(D),
         "pos": rn(P, D), "prefix": rn(max(spec.prefix_tokens, 1), D), "ln_w": torch.ones(D, device=dev, dtype=dt),
         "ln_b": torch.zeros(D, device=dev, dtype=dt), "layers": []}
    for _ in range(spec.layers + (0 if spec.final_ln else 1)):  # the reference also runs the discarded last block
        L = {"qkv_w": rn(3 * D, D, std=D**-0.5), "qkv_b": rn(3 * D), "proj_w": rn(D, D, std=D**-0.5), "proj_b": rn(D),
             "fc1_w": rn(Mh, D, std=D**-0.5), "fc1_b": rn(Mh), "fc2_w": rn(D, Mh, std=Mh**-0.5), "fc2_b": rn(D), "ls": rn(D, std=1.0)}
        if spec.temporal_frames:
            L.update({"t_qkv_w": rn(3 * D, D, std=D**-0.5), "t_qkv_b": rn(3 * D), "t_proj_w": rn(D, D, std=D**-0.5),
                      "t_proj_b": rn(D), "t_emb": rn(spec.temporal_frames, D)})
        W["layers"].append(L)
    return W


def mhsa(x, qkv_w, qkv_b, proj_w, proj_b, heads):
    N, L, D = x.shape
    qkv = F.linear(x, qkv_w, qkv_b).reshape(N, L, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
    o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
    return F.linear(o.transpose(1, 2).reshape(N, L, D), proj_w, proj_b)


def encoder(pix, spec, W):
    Bv = pix.shape[0]
    x = pix if spec.pix_layout == "BFCHW" else pix.permute(0, 2, 1, 3, 4)
    if spec.tubelet == 1:
        e = F.conv2d(x.reshape(-1, 3, spec.img, spec.img), W["patch_w"], W["patch_b"], stride=spec.patch).flatten(2).transpose(1, 2)
    else:
        e = F.conv3d(x.permute(0, 2, 1, 3, 4), W["patch_w"], W["patch_b"], stride=(spec.tubelet, spec.patch, spec.patch)).flatten(2).transpose(1, 2)
    e = e + W["pos"]
    if spec.prefix_tokens:
        e = torch.cat([W["prefix"][: spec.prefix_tokens].expand(e.shape[0], -1, -1), e], 1)
    x = e
    D = spec.dim
    for L in W["layers"]:
        if spec.temporal_frames:
            t = spec.temporal_frames
            bt, n, d = x.shape
            h = x.reshape(bt // t, t, n, d).permute(0, 2, 1, 3).reshape(-1, t, d) + L["t_emb"]
            x = h.reshape(bt // t, n, t, d).permute(0, 2, 1, 3).reshape(bt, n, d)
            h = F.layer_norm(x.reshape(bt // t, t, n, d).permute(0, 2, 1, 3).reshape(-1, t, d), (D,), W["ln_w"], W["ln_b"])
            h = mhsa(h, L["t_qkv_w"], L["t_qkv_b"], L["t_proj_w"], L["t_proj_b"], spec.heads)
            x = x + h.reshape(bt // t, n, t, d).permute(0, 2, 1, 3).reshape(bt, n, d)
        h = mhsa(F.layer_norm(x, (D,), W["ln_w"], W["ln_b"]), L["qkv_w"], L["qkv_b"], L["proj_w"], L["proj_b"], spec.heads)
        x = x + (h * L["ls"] if spec.layerscale else h)
        h = F.linear(F.gelu(F.linear(F.layer_norm(x, (D,), W["ln_w"], W["ln_b"]), L["fc1_w"], L["fc1_b"]),
                            approximate="tanh" if spec.act == "gelu_tanh" else "none"), L["fc2_w"], L["fc2_b"])
        x = x + (h * L["ls"] if spec.layerscale else h)
    if spec.final_ln:
        x = F.layer_norm(x, (D,), W["ln_w"], W["ln_b"])
    return x[:, spec.prefix_tokens:].reshape(Bv, -1, D)


specs = merv_full_specs()
Ws = [make(s) for s in specs]
projs = [(rn(4096, s.dim, std=s.dim**-0.5), rn(4096)) for s in specs]
pix = [torch.randn(s.pixel_shape(B), generator=g, device=dev).to(dt) for s in specs]
u = rn(4096, std=1.0).float()


@torch.no_grad()
def step():
    outs = []
    for s, W, (pw, pb), p in zip(specs, Ws, projs, pix):
        tok = encoder(p, s, W)
        x = tok.reshape(B, s.t_out, s.hp, s.hp, s.dim).permute(0, 4, 1, 2, 3)
        x = F.adaptive_avg_pool3d(x, (s.t_out, 8, 8)).permute(0, 2, 3, 4, 1).reshape(B, -1, s.dim)
        outs.append(F.linear(x, pw, pb))
    V = torch.stack(outs, 1)
    w = (V.float().mean(2) @ u).softmax(-1)
    return torch.einsum("be,betc->btc", w.to(dt), V)


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(json.dumps({"what": "PyTorch-ROCm eager bf16 (library GEMM + SDPA), reference-style sequential encoders, all blocks the reference runs",
                  "videos_per_step": B, "ms_per_step": round(el / steps * 1e3, 2), "visual_tokens_per_s": round(B * 1024 * steps / el, 1),
                  "torch": torch.__version__}))
