#!/usr/bin/env python3
"""What the reference's own stack does on this GPU: the same four encoders + projectors + fusion written with plain
PyTorch-ROCm ops in bf16 (F.linear -> hipBLASLt/rocBLAS, F.scaled_dot_product_attention, F.layer_norm, F.gelu), i.e. the
kernels timm / transformers launch under `vidlm.to(torch.bfloat16)` (scripts/quick_start.py:12) + autocast
(merv.py:816), encoders run one after the other (merv.py:563-566). Measurement / calibration only -- nothing here is part
of the product, and nothing in merv_amd/ imports it.

Importable pieces (tools/parity_calibration.py uses them to calibrate the bf16 tolerance against the reference stack's own
numerics): `to_ref_stack`, `encoder_bf16`, `path_bf16`. Run as a script it times the stack and prints one JSON line."""
import dataclasses
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.nn.functional as F

dt = torch.bfloat16


def to_ref_stack(W, device):
    """Canonical weight dict (merv_amd/encoder.py) -> every parameter bf16 on `device`: what `vidlm.to(torch.bfloat16)`
    leaves the reference's modules with (LayerNorm parameters and biases included)."""
    out = {k: v.to(device=device, dtype=dt) for k, v in W.items() if k != "layers"}
    out["layers"] = [{k: v.to(device=device, dtype=dt) for k, v in L.items()} for L in W["layers"]]
    return out


def _mhsa(x, qkv_w, qkv_b, proj_w, proj_b, heads):
    N, L, D = x.shape
    qkv = F.linear(x, qkv_w, qkv_b).reshape(N, L, 3, heads, D // heads).permute(2, 0, 3, 1, 4)
    o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])  # timm fused_attn / HF sdpa attention
    return F.linear(o.transpose(1, 2).reshape(N, L, D), proj_w, proj_b)


def _ln(x, w, b, eps):
    # under autocast layer_norm runs in fp32 and its output is cast back by the next bf16 op
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps).to(dt)


def _act(name, x):
    if name == "gelu_erf":
        return F.gelu(x)
    if name == "gelu_tanh":  # HF gelu_fast
        return 0.5 * x * (1.0 + torch.tanh(x * 0.7978845608 * (1.0 + 0.044715 * x * x)))
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    raise ValueError(name)


@torch.no_grad()
def encoder_bf16(pix, spec, W, layers=None):
    """pix in the spec's layout (bf16, device) -> [B, num_patches, D] bf16 with torch library kernels.
    `layers`: blocks to run (default spec.layers = the consumed ones; the reference runs one more and discards it)."""
    Bv = pix.shape[0]
    D = spec.dim
    x = pix.to(dt)
    x = x if spec.pix_layout == "BFCHW" else x.permute(0, 2, 1, 3, 4)
    pb = W.get("patch_b")
    if spec.tubelet == 1:
        e = F.conv2d(x.reshape(-1, 3, spec.img, spec.img), W["patch_w"].reshape(D, 3, spec.patch, spec.patch), pb, stride=spec.patch)
    else:
        e = F.conv3d(x.permute(0, 2, 1, 3, 4), W["patch_w"].reshape(D, 3, spec.tubelet, spec.patch, spec.patch), pb,
                     stride=(spec.tubelet, spec.patch, spec.patch))
    e = e.flatten(2).transpose(1, 2) + W["pos"]
    if spec.prefix_tokens:
        e = torch.cat([W["prefix"][None].expand(e.shape[0], -1, -1), e], 1)
    if spec.pre_ln:
        e = _ln(e, W["pre_ln_w"], W["pre_ln_b"], spec.ln_eps)
    x = e
    n_run = spec.layers if layers is None else layers
    for li in range(n_run):
        L = W["layers"][li % len(W["layers"])]
        if spec.temporal_frames:
            t = spec.temporal_frames
            bt, n, d = x.shape
            h = x.reshape(bt // t, t, n, d).permute(0, 2, 1, 3).reshape(-1, t, d) + L["t_emb"][None, :t]
            x = h.reshape(bt // t, n, t, d).permute(0, 2, 1, 3).reshape(bt, n, d)
            h = _ln(x.reshape(bt // t, t, n, d).permute(0, 2, 1, 3).reshape(-1, t, d), L["t_ln_w"], L["t_ln_b"], spec.ln_eps)
            h = _mhsa(h, L["t_qkv_w"], L["t_qkv_b"], L["t_proj_w"], L["t_proj_b"], spec.heads)
            x = x + h.reshape(bt // t, n, t, d).permute(0, 2, 1, 3).reshape(bt, n, d)
        h = _mhsa(_ln(x, L["ln1_w"], L["ln1_b"], spec.ln_eps), L["qkv_w"], L["qkv_b"], L["proj_w"], L["proj_b"], spec.heads)
        x = x + (h * L["ls1"] if spec.layerscale else h)
        h = F.linear(_act(spec.act, F.linear(_ln(x, L["ln2_w"], L["ln2_b"], spec.ln_eps), L["fc1_w"], L["fc1_b"])), L["fc2_w"], L["fc2_b"])
        x = x + (h * L["ls2"] if spec.layerscale else h)
        if li == spec.layers - 1:
            keep = x  # the consumed output; later blocks are the ones the reference computes and discards
    x = keep if n_run >= spec.layers else x
    if spec.final_ln:
        x = _ln(x, W["final_ln_w"], W["final_ln_b"], spec.ln_eps)
    return x[:, spec.prefix_tokens:].reshape(Bv, -1, D)


@torch.no_grad()
def path_bf16(pixels, specs, Ws, projs, u, extra_block=False):
    """Encoders one after the other -> AdaptiveAvgPool3d + Linear -> folded-query fusion. Returns (fused, weights, projected)."""
    outs = []
    for s, W, (pw, pb), p in zip(specs, Ws, projs, pixels):
        tok = encoder_bf16(p, s, W, layers=s.layers + (1 if extra_block and not s.final_ln else 0))
        B = tok.shape[0]
        x = tok.reshape(B, s.t_out, s.hp, s.hp, s.dim).permute(0, 4, 1, 2, 3)
        x = F.adaptive_avg_pool3d(x, (s.t_out, 8, 8)).permute(0, 2, 3, 4, 1).reshape(B, -1, s.dim)
        outs.append(F.linear(x, pw.to(dt), pb.to(dt)))
    V = torch.stack(outs, 1)
    w = (V.float().mean(2) @ u.float()).softmax(-1)
    return torch.einsum("be,betc->btc", w.to(dt), V), w, outs


def main():
    from merv_amd.backbones import random_weights
    from merv_amd.encoder import merv_full_specs
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    specs = merv_full_specs()
    # two blocks of weights per encoder, reused cyclically (timing does not depend on the values)
    Ws = [to_ref_stack(random_weights(dataclasses.replace(s, layers=2), seed=i, device=dev), dev) for i, s in enumerate(specs)]
    g = torch.Generator(device=dev).manual_seed(0)
    projs = [(torch.randn(4096, s.dim, generator=g, device=dev) * s.dim**-0.5, torch.randn(4096, generator=g, device=dev) * 0.02) for s in specs]
    pix = [torch.randn(s.pixel_shape(B), generator=g, device=dev).to(dt) for s in specs]
    u = torch.randn(4096, generator=g, device=dev)

    def step():
        return path_bf16(pix, specs, Ws, projs, u, extra_block=True)[0]

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"what": "PyTorch-ROCm eager bf16 (library GEMM + SDPA), reference-style sequential encoders, all blocks the reference runs",
                      "videos_per_step": B, "ms_per_step": round(el / steps * 1e3, 2), "visual_tokens_per_s": round(B * 1024 * steps / el, 1),
                      "torch": torch.__version__}))


if __name__ == "__main__":
    main()
