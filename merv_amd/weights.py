"""
Weight ingestion: upstream state dicts -> the canonical per-encoder weight dict consumed by HipEncoder
(merv_amd/encoder.py) -- so the runtime needs neither timm nor transformers model classes.

Sources ("State-dict keys observed"):
  * LanguageBind vision tower  (languagebind/video/modeling_video.py: CLIPVisionTransformer; hub prefix `vision_model.`)
  * HF VivitModel              (transformers >= 5 names `layers.N.attention.q_proj`, and the 4.44.2 / hub names
                                `vivit.encoder.layer.N.attention.attention.query` ...)
  * timm VisionTransformer     (DINOv2 reg4 / SigLIP: `blocks.N.attn.qkv`, `cls_token`, `reg_token`, `pos_embed`,
                                `blocks.N.ls1.gamma`) -- timm is not installed here: semantics from timm 0.9.10,
                                "timm parity unpinned" (cross-checked against the HF equivalents below)
  * HF Dinov2WithRegistersModel / SiglipVisionModel (used as the cross-check of the timm semantics)

Canonical form: see merv_amd/encoder.py docstring. All tensors fp32 CPU.
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional

import torch
import torch.nn.functional as F


def _strip(sd: Mapping[str, torch.Tensor], prefixes) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        for p in prefixes:
            if k.startswith(p):
                k = k[len(p):]
                break
        out[k] = torch.as_tensor(v).float()
    return out


def _cat_qkv(sd, fmt: str):
    w = torch.cat([sd[fmt.format(p=p) + ".weight"] for p in "qkv"], 0)
    b = torch.cat([sd[fmt.format(p=p) + ".bias"] for p in "qkv"], 0)
    return w, b


def from_languagebind_vision(sd: Mapping[str, torch.Tensor], n_layers: Optional[int] = None) -> Dict:
    """CLIP ViT with a temporal-attention sub-block per layer. cls row = class_embedding + position_embedding[0]
    (HF CLIPVisionEmbeddings adds the position embedding to every token, incl. cls); conv has no bias."""
    sd = _strip(sd, ("vision_model.",))
    pos = sd["embeddings.position_embedding.weight"]
    W = {
        "patch_w": sd["embeddings.patch_embedding.weight"].flatten(1),
        "prefix": (sd["embeddings.class_embedding"] + pos[0])[None],
        "pos": pos[1:],
        "pre_ln_w": sd["pre_layrnorm.weight"], "pre_ln_b": sd["pre_layrnorm.bias"],
        "layers": [],
    }
    total = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
    for i in range(total if n_layers is None else n_layers):
        p = f"encoder.layers.{i}."
        qkv_w, qkv_b = _cat_qkv(sd, p + "self_attn.{p}_proj")
        Lw = {
            "ln1_w": sd[p + "layer_norm1.weight"], "ln1_b": sd[p + "layer_norm1.bias"],
            "qkv_w": qkv_w, "qkv_b": qkv_b,
            "proj_w": sd[p + "self_attn.out_proj.weight"], "proj_b": sd[p + "self_attn.out_proj.bias"],
            "ln2_w": sd[p + "layer_norm2.weight"], "ln2_b": sd[p + "layer_norm2.bias"],
            "fc1_w": sd[p + "mlp.fc1.weight"], "fc1_b": sd[p + "mlp.fc1.bias"],
            "fc2_w": sd[p + "mlp.fc2.weight"], "fc2_b": sd[p + "mlp.fc2.bias"],
        }
        if p + "temporal_embedding" in sd:
            t_w, t_b = _cat_qkv(sd, p + "temporal_attn.{p}_proj")
            Lw.update({
                "t_emb": sd[p + "temporal_embedding"][0],
                "t_ln_w": sd[p + "temporal_layer_norm1.weight"], "t_ln_b": sd[p + "temporal_layer_norm1.bias"],
                "t_qkv_w": t_w, "t_qkv_b": t_b,
                "t_proj_w": sd[p + "temporal_attn.out_proj.weight"], "t_proj_b": sd[p + "temporal_attn.out_proj.bias"],
            })
        W["layers"].append(Lw)
    return W


def from_hf_vivit(sd: Mapping[str, torch.Tensor], n_layers: Optional[int] = None) -> Dict:
    """HF VivitModel. Accepts transformers-5 names and the 4.44.2 / hub-checkpoint names."""
    sd = _strip(sd, ("vivit.",))
    new = any(k.startswith("layers.") for k in sd)
    pos = sd["embeddings.position_embeddings"][0]
    W = {
        "patch_w": sd["embeddings.patch_embeddings.projection.weight"].flatten(1),
        "patch_b": sd["embeddings.patch_embeddings.projection.bias"],
        "prefix": (sd["embeddings.cls_token"][0, 0] + pos[0])[None],
        "pos": pos[1:],
        "final_ln_w": sd["layernorm.weight"], "final_ln_b": sd["layernorm.bias"],
        "layers": [],
    }
    lp = "layers." if new else "encoder.layer."
    total = 1 + max(int(k[len(lp):].split(".")[0]) for k in sd if k.startswith(lp))
    for i in range(total if n_layers is None else n_layers):
        p = f"{lp}{i}."
        if new:
            qkv_w, qkv_b = _cat_qkv(sd, p + "attention.{p}_proj")
            ow, ob = sd[p + "attention.o_proj.weight"], sd[p + "attention.o_proj.bias"]
            f1w, f1b, f2w, f2b = (sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], sd[p + "mlp.fc2.weight"],
                                  sd[p + "mlp.fc2.bias"])
        else:
            names = {"q": "query", "k": "key", "v": "value"}
            qkv_w = torch.cat([sd[p + f"attention.attention.{names[c]}.weight"] for c in "qkv"], 0)
            qkv_b = torch.cat([sd[p + f"attention.attention.{names[c]}.bias"] for c in "qkv"], 0)
            ow, ob = sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"]
            f1w, f1b = sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]
            f2w, f2b = sd[p + "output.dense.weight"], sd[p + "output.dense.bias"]
        W["layers"].append({
            "ln1_w": sd[p + "layernorm_before.weight"], "ln1_b": sd[p + "layernorm_before.bias"],
            "qkv_w": qkv_w, "qkv_b": qkv_b, "proj_w": ow, "proj_b": ob,
            "ln2_w": sd[p + "layernorm_after.weight"], "ln2_b": sd[p + "layernorm_after.bias"],
            "fc1_w": f1w, "fc1_b": f1b, "fc2_w": f2w, "fc2_b": f2b,
        })
    return W


def resample_abs_pos_embed(pos: torch.Tensor, new_hw: int) -> torch.Tensor:
    """timm.layers.resample_abs_pos_embed for a square grid without prefix tokens: bicubic, antialias=True,
    align_corners=False (what timm applies at load when img_size != the checkpoint's; 37x37 -> 16x16 for DINOv2@224)."""
    P, D = pos.shape
    old = int(round(P**0.5))
    if old == new_hw:
        return pos
    g = pos.reshape(1, old, old, D).permute(0, 3, 1, 2).float()
    g = F.interpolate(g, size=(new_hw, new_hw), mode="bicubic", antialias=True, align_corners=False)
    return g.permute(0, 2, 3, 1).reshape(new_hw * new_hw, D)


def from_timm_vit(sd: Mapping[str, torch.Tensor], n_layers: Optional[int] = None, grid: Optional[int] = None) -> Dict:
    """timm VisionTransformer (vit_large_patch14_reg4_dinov2 / vit_base_patch16_siglip_224).
    DINOv2-reg (no_embed_class): pos_embed covers patches only; cls_token already holds cls + pos[0] (timm's
    checkpoint filter folds it); register tokens carry no position."""
    sd = _strip(sd, ())
    pos = sd["pos_embed"][0]
    pre = []
    if "cls_token" in sd:
        pre.append(sd["cls_token"][0])
    if "reg_token" in sd:
        pre.append(sd["reg_token"][0])
    npre = sum(p.shape[0] for p in pre)
    P_total = pos.shape[0]
    root = int(round(P_total**0.5))
    if root * root != P_total:  # pos_embed includes prefix positions (plain ViT, no_embed_class=False)
        pre_t = torch.cat(pre, 0) + pos[:npre]
        pos = pos[npre:]
        pre = [pre_t]
    if grid is not None:
        pos = resample_abs_pos_embed(pos, grid)
    W = {
        "patch_w": sd["patch_embed.proj.weight"].flatten(1), "patch_b": sd["patch_embed.proj.bias"],
        "pos": pos, "layers": [],
    }
    if pre:
        W["prefix"] = torch.cat(pre, 0)
    total = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(total if n_layers is None else n_layers):
        p = f"blocks.{i}."
        Lw = {
            "ln1_w": sd[p + "norm1.weight"], "ln1_b": sd[p + "norm1.bias"],
            "qkv_w": sd[p + "attn.qkv.weight"], "qkv_b": sd[p + "attn.qkv.bias"],
            "proj_w": sd[p + "attn.proj.weight"], "proj_b": sd[p + "attn.proj.bias"],
            "ln2_w": sd[p + "norm2.weight"], "ln2_b": sd[p + "norm2.bias"],
            "fc1_w": sd[p + "mlp.fc1.weight"], "fc1_b": sd[p + "mlp.fc1.bias"],
            "fc2_w": sd[p + "mlp.fc2.weight"], "fc2_b": sd[p + "mlp.fc2.bias"],
        }
        if p + "ls1.gamma" in sd:
            Lw["ls1"], Lw["ls2"] = sd[p + "ls1.gamma"], sd[p + "ls2.gamma"]
        W["layers"].append(Lw)
    if "norm.weight" in sd:  # timm's final norm: only the bare `dinov2-video` id applies it (forward_features + token pool)
        W["final_ln_w"], W["final_ln_b"] = sd["norm.weight"], sd["norm.bias"]
    return W


def from_timm_attn_pool(sd: Mapping[str, torch.Tensor]) -> Dict:
    """timm AttentionPoolLatent (`attn_pool.*` of vit_base_patch16_siglip_224, global_pool='map') -> the MAP-head dict."""
    sd = _strip(sd, ())
    p = "attn_pool."
    return {"latent": sd[p + "latent"].reshape(1, -1), "q_w": sd[p + "q.weight"], "q_b": sd[p + "q.bias"],
            "kv_w": sd[p + "kv.weight"], "kv_b": sd[p + "kv.bias"], "proj_w": sd[p + "proj.weight"], "proj_b": sd[p + "proj.bias"],
            "norm_w": sd[p + "norm.weight"], "norm_b": sd[p + "norm.bias"], "fc1_w": sd[p + "mlp.fc1.weight"],
            "fc1_b": sd[p + "mlp.fc1.bias"], "fc2_w": sd[p + "mlp.fc2.weight"], "fc2_b": sd[p + "mlp.fc2.bias"]}


def from_hf_siglip_head(sd: Mapping[str, torch.Tensor]) -> Dict:
    """transformers SiglipMultiheadAttentionPoolingHead (`head.*`): nn.MultiheadAttention's packed in_proj = [q; k; v] rows."""
    g = lambda k: next(v for n, v in sd.items() if n.endswith("head." + k) or n == k)
    D = g("probe").shape[-1]
    w, b = g("attention.in_proj_weight"), g("attention.in_proj_bias")
    return {"latent": g("probe").reshape(1, D), "q_w": w[:D], "q_b": b[:D], "kv_w": w[D:], "kv_b": b[D:],
            "proj_w": g("attention.out_proj.weight"), "proj_b": g("attention.out_proj.bias"), "norm_w": g("layernorm.weight"),
            "norm_b": g("layernorm.bias"), "fc1_w": g("mlp.fc1.weight"), "fc1_b": g("mlp.fc1.bias"), "fc2_w": g("mlp.fc2.weight"),
            "fc2_b": g("mlp.fc2.bias")}


def from_hf_dinov2(sd: Mapping[str, torch.Tensor], n_layers: Optional[int] = None) -> Dict:
    """HF Dinov2WithRegistersModel: emb = cat(cls, patches) + pos, registers inserted after cls without position."""
    sd = _strip(sd, ())
    pos = sd["embeddings.position_embeddings"][0]
    W = {
        "patch_w": sd["embeddings.patch_embeddings.projection.weight"].flatten(1),
        "patch_b": sd["embeddings.patch_embeddings.projection.bias"],
        "prefix": torch.cat([sd["embeddings.cls_token"][0] + pos[:1], sd["embeddings.register_tokens"][0]], 0),
        "pos": pos[1:], "layers": [],
    }
    total = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layer."))
    names = {"q": "query", "k": "key", "v": "value"}
    for i in range(total if n_layers is None else n_layers):
        p = f"encoder.layer.{i}."
        W["layers"].append({
            "ln1_w": sd[p + "norm1.weight"], "ln1_b": sd[p + "norm1.bias"],
            "qkv_w": torch.cat([sd[p + f"attention.attention.{names[c]}.weight"] for c in "qkv"], 0),
            "qkv_b": torch.cat([sd[p + f"attention.attention.{names[c]}.bias"] for c in "qkv"], 0),
            "proj_w": sd[p + "attention.output.dense.weight"], "proj_b": sd[p + "attention.output.dense.bias"],
            "ls1": sd[p + "layer_scale1.lambda1"],
            "ln2_w": sd[p + "norm2.weight"], "ln2_b": sd[p + "norm2.bias"],
            "fc1_w": sd[p + "mlp.fc1.weight"], "fc1_b": sd[p + "mlp.fc1.bias"],
            "fc2_w": sd[p + "mlp.fc2.weight"], "fc2_b": sd[p + "mlp.fc2.bias"],
            "ls2": sd[p + "layer_scale2.lambda1"],
        })
    return W


def from_hf_siglip(sd: Mapping[str, torch.Tensor], n_layers: Optional[int] = None) -> Dict:
    sd = _strip(sd, ("vision_model.",))
    W = {
        "patch_w": sd["embeddings.patch_embedding.weight"].flatten(1), "patch_b": sd["embeddings.patch_embedding.bias"],
        "pos": sd["embeddings.position_embedding.weight"], "layers": [],
    }
    total = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
    for i in range(total if n_layers is None else n_layers):
        p = f"encoder.layers.{i}."
        qkv_w, qkv_b = _cat_qkv(sd, p + "self_attn.{p}_proj")
        W["layers"].append({
            "ln1_w": sd[p + "layer_norm1.weight"], "ln1_b": sd[p + "layer_norm1.bias"],
            "qkv_w": qkv_w, "qkv_b": qkv_b,
            "proj_w": sd[p + "self_attn.out_proj.weight"], "proj_b": sd[p + "self_attn.out_proj.bias"],
            "ln2_w": sd[p + "layer_norm2.weight"], "ln2_b": sd[p + "layer_norm2.bias"],
            "fc1_w": sd[p + "mlp.fc1.weight"], "fc1_b": sd[p + "mlp.fc1.bias"],
            "fc2_w": sd[p + "mlp.fc2.weight"], "fc2_b": sd[p + "mlp.fc2.bias"],
        })
    return W


def to_timm_names(W: Dict, no_embed_class: bool = True) -> Dict[str, torch.Tensor]:
    """Inverse of from_timm_vit for tests (canonical -> timm-style state dict)."""
    sd = {"patch_embed.proj.weight": W["patch_w"], "patch_embed.proj.bias": W["patch_b"], "pos_embed": W["pos"][None]}
    if "prefix" in W:
        sd["cls_token"] = W["prefix"][:1][None]
        if W["prefix"].shape[0] > 1:
            sd["reg_token"] = W["prefix"][1:][None]
    for i, Lw in enumerate(W["layers"]):
        p = f"blocks.{i}."
        sd.update({p + "norm1.weight": Lw["ln1_w"], p + "norm1.bias": Lw["ln1_b"], p + "attn.qkv.weight": Lw["qkv_w"],
                   p + "attn.qkv.bias": Lw["qkv_b"], p + "attn.proj.weight": Lw["proj_w"], p + "attn.proj.bias": Lw["proj_b"],
                   p + "norm2.weight": Lw["ln2_w"], p + "norm2.bias": Lw["ln2_b"], p + "mlp.fc1.weight": Lw["fc1_w"],
                   p + "mlp.fc1.bias": Lw["fc1_b"], p + "mlp.fc2.weight": Lw["fc2_w"], p + "mlp.fc2.bias": Lw["fc2_b"]})
        if "ls1" in Lw:
            sd[p + "ls1.gamma"], sd[p + "ls2.gamma"] = Lw["ls1"], Lw["ls2"]
    return sd
