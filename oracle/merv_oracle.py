"""
merv_oracle.py -- CPU restatement (torch fp32 / pure Python float64) of the reference's multi-encoder video
forward path.  *** TEST INFRASTRUCTURE ONLY ***: it may be imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg, as the checker / reported baseline -- never by the product path (merv_amd/), which
fails loudly when the HIP library is missing.

Every function names the reference code (path:line under /root/reference) it restates.

Pinning status (see DESIGN.md "Oracle"):
  * frame_indices        -- pinned against numpy.linspace goldens (tests/golden/frame_indices.json), incl. the
                            reference's own eval_data/dummy_mcq end_frame=595 case.
  * projector / fusion   -- pinned against the reference's own classes (merv/util/nn_utils.py imported by path;
                            tests/golden/projector_fusion.npz).
  * LanguageBind encoder -- pinned against the reference's vendored CLIPVisionTransformer
                            (languagebind/video/modeling_video.py imported by path; tests/golden/languagebind.npz).
  * ViViT encoder        -- pinned against transformers.VivitModel, the class the reference calls (vivit.py:42).
  * DINOv2 / SigLIP      -- timm 0.9.10 is not installed here and not vendored by the reference: restated from
                            timm's published VisionTransformer semantics and cross-checked against HF
                            Dinov2WithRegistersModel / SiglipVisionModel. "timm parity unpinned".
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------------------
# a1. frame-index selection -- merv/preprocessing/datasets/datasets.py:131-141 (+ NaN guards :46-52)
# ------------------------------------------------------------------------------------------------------------
def _linspace_int(start: float, stop: float, num: int) -> List[int]:
    """numpy.linspace(start, stop, num, dtype=int): float64 ramp k*step+start (two roundings), last sample forced to
    `stop`, floor, cast (numpy/_core/function_base.py). Pure Python floats are IEEE double, like numpy's."""
    if num <= 0:
        return []
    div = num - 1
    delta = stop - start
    out = []
    if div > 0:
        step = delta / div
        for k in range(num):
            if step == 0:
                y = (k / div) * delta
            else:
                y = k * step
            y = y + start
            if k == num - 1:
                y = stop
            out.append(int(math.floor(y)))
    else:
        out.append(int(math.floor(0.0 * delta + start)))
    return out


def frame_indices(
    video_num_frames: int,
    avg_fps: float,
    clip_start_sec: Optional[float] = 0.0,
    clip_end_sec: Optional[float] = None,
    num_frames: int = 8,
    end_frame: Optional[int] = None,
) -> List[int]:
    """datasets.py:46-52 (NaN guards), :126-141 (index math of the decord branch)."""
    if clip_start_sec is not None and math.isnan(clip_start_sec):
        clip_start_sec = 0
    if clip_end_sec is not None and math.isnan(clip_end_sec):
        clip_end_sec = None
    total_secs = video_num_frames / avg_fps  # :128
    if end_frame is None or end_frame < 0:  # :131
        if clip_end_sec is None:
            clip_end_sec = total_secs
        start = clip_start_sec * avg_fps
        stop = min(video_num_frames - 1, clip_end_sec * avg_fps - 1)  # :135-136
    else:
        start = 0
        stop = min(video_num_frames - 1, end_frame)  # :139-140
    return _linspace_int(float(start), float(stop), num_frames)


def temporal_subsample(loaded_frames: int, max_nf: int, nf: int) -> List[int]:
    """merv/models/vidlms/merv.py:803-806: video[:: max(num_frames) // nf] (over-samples when nf does not divide)."""
    return list(range(loaded_frames))[:: max_nf // nf]


# ------------------------------------------------------------------------------------------------------------
# a4-a7. the four encoders as one parameterised pre-LN ViT
# ------------------------------------------------------------------------------------------------------------
@dataclass
class EncoderCfg:
    name: str
    dim: int
    heads: int
    mlp_dim: int
    layers: int  # blocks actually consumed
    patch: int
    tubelet: int
    img: int
    frames: int
    pix_layout: str  # "BFCHW" | "BCFHW"
    prefix_tokens: int
    joint_space_time: bool
    pre_ln: bool
    final_ln: bool
    layerscale: bool
    temporal_frames: int
    act: str  # "gelu_erf" | "gelu_tanh" | "quick_gelu"
    ln_eps: float

    @property
    def hp(self) -> int:
        return self.img // self.patch

    @property
    def t_out(self) -> int:
        return self.frames // self.tubelet

    @property
    def s_out(self) -> int:
        return self.hp * self.hp


def merv_full_cfgs(depth_scale: float = 1.0) -> List[EncoderCfg]:
    """merv-full encoder list in registry order (merv/conf/models.py:106-113), frames [16,16,32,16] (:118).
    Layer counts = blocks whose output is consumed."""

    def L(n):
        return max(1, int(round(n * depth_scale)))

    return [
        # languagebind-video-noclass: CLIP ViT-L/14 + temporal attention, hidden_states[-2] (languagebind/__init__.py:85)
        EncoderCfg("languagebind", 1024, 16, 4096, L(23), 14, 1, 224, 16, "BCFHW", 1, False, True, False, False, 8,
                   "gelu_erf", 1e-5),
        # dinov2-video-all-tokens: timm vit_large_patch14_reg4_dinov2, get_intermediate_layers(n={22}) (dinov2_video.py:63-66)
        EncoderCfg("dinov2", 1024, 16, 4096, L(23), 14, 1, 224, 16, "BFCHW", 5, False, False, False, True, 0,
                   "gelu_erf", 1e-6),
        # vivit-google-b-all-no-cls-16frames: HF VivitModel, last_hidden_state (final LN) (vivit.py:104-114)
        EncoderCfg("vivit", 768, 12, 3072, L(12), 16, 2, 224, 32, "BFCHW", 1, True, False, True, False, 0,
                   "gelu_tanh", 1e-6),
        # siglip-vit-b16-224px-all-no-cls: timm vit_base_patch16_siglip_224, n={10} (siglip.py:60-63)
        EncoderCfg("siglip", 768, 12, 3072, L(11), 16, 1, 224, 16, "BFCHW", 0, False, False, False, False, 0,
                   "gelu_erf", 1e-6),
    ]


def act_fn(name: str, x: torch.Tensor) -> torch.Tensor:
    if name == "gelu_erf":  # timm nn.GELU
        return F.gelu(x)
    if name == "gelu_tanh":  # HF "gelu_fast" (VivitConfig.hidden_act)
        return 0.5 * x * (1.0 + torch.tanh(x * 0.7978845608 * (1.0 + 0.044715 * x * x)))
    if name == "quick_gelu":  # CLIP default (configuration_video.py:190)
        return x * torch.sigmoid(1.702 * x)
    raise ValueError(name)


def mx_quantize_dequantize(x: torch.Tensor) -> torch.Tensor:
    """OCP Microscaling FP8 (e4m3, 32-element blocks along the last dim, E8M0 shared exponent floor(log2 amax) - 8,
    plus one when the scaled block maximum would exceed 448; round-to-nearest-even) applied and undone: the values the MXFP8 mode's GEMMs multiply. No counterpart in
    the reference (BASELINE.json configs[4] is this build's fp8 mode); used only to check that mode."""
    shp = x.shape
    v = x.float().reshape(-1, shp[-1] // 32, 32)
    amax = v.abs().amax(-1)
    _, ex = torch.frexp(amax)
    e = ex - 1 - 8
    e = e + ((amax / torch.exp2(e.float())) > 448).to(e.dtype)  # smallest power-of-two scale under which nothing saturates
    e = e.clamp(min=-127)
    e = torch.where(amax == 0, torch.full_like(e, -127), e)
    sc = torch.exp2(e.float())[..., None]
    q = (v / sc).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    return (q * sc).reshape(shp)


def mx_linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """F.linear on MXFP8-quantised input and weight (the input first rounded to bf16, as the HIP path stores it)."""
    return F.linear(mx_quantize_dequantize(x.to(torch.bfloat16).float()), mx_quantize_dequantize(w.to(torch.bfloat16).float()), b)


def mx_folded_ln_linear(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], eps: float,
                        scale_rows: int = 0, row_scale: float = 1.0) -> torch.Tensor:
    """linear(LayerNorm(x)) as the MXFP8 mode computes it under the LayerNorm fold (merv_amd/csrc/capi.hip, round 6): the RAW stream x
    (bf16) and the folded weight W' = bf16(W * gamma) are what is quantised; the normalisation is algebra behind the product --
    y = rstd * (dq(x) . dq(W')^T) - rstd * mean * colsum(dq(W')) + (W . beta + b), statistics of the bf16 stream. `scale_rows` /
    `row_scale`: the first rows of W' (the q rows of qkv) carry a factor before rounding, as the folded weight does (the attention's
    scale * log2 e); it is divided out of those outputs again here, so the result is the plain linear's."""
    xb = x.to(torch.bfloat16).float()
    mean = xb.mean(-1, keepdim=True)
    var = ((xb - mean) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    rs = torch.ones(w.shape[0])
    rs[:scale_rows] = row_scale
    wb = w.to(torch.bfloat16).float()
    wf = mx_quantize_dequantize((wb * gamma[None] * rs[:, None]).to(torch.bfloat16).float())
    db = (wb @ beta + (b if b is not None else 0.0)) * rs
    y = rstd * (mx_quantize_dequantize(xb) @ wf.t()) - rstd * mean * wf.sum(-1) + db
    return y / rs


def _mhsa(x: torch.Tensor, qkv_w, qkv_b, proj_w, proj_b, heads: int, linear=F.linear, qkv_linear=None) -> torch.Tensor:
    """softmax(q k^T / sqrt(d)) v with fused qkv weights; x [N, L, D].
    timm Attention.forward; HF CLIPAttention (q*scale, softmax, no mask; modeling_video.py:98,168); HF VivitSelfAttention."""
    N, L, D = x.shape
    hd = D // heads
    qkv = (qkv_linear or linear)(x, qkv_w, qkv_b).reshape(N, L, 3, heads, hd).permute(2, 0, 3, 1, 4)  # (qkv_linear: the input projection alone in another form)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q @ k.transpose(-1, -2)) * (hd**-0.5)
    att = att.softmax(dim=-1)
    o = (att @ v).transpose(1, 2).reshape(N, L, D)
    return linear(o, proj_w, proj_b)


def encoder_embed(pix: torch.Tensor, cfg: EncoderCfg, W: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Patch / tubelet embedding + prefix tokens + position embedding -> [nseq, ntok, D].
    timm PatchEmbed + VisionTransformer._pos_embed (no_embed_class for DINOv2: pos added to patches only, then
    cls/reg concatenated); HF CLIPVisionEmbeddings (modeling_video.py:682); HF VivitEmbeddings."""
    B = pix.shape[0]
    x = pix.float()
    if cfg.pix_layout == "BCFHW":  # LanguageBind: rearrange "b c t h w -> (b t) c h w" (modeling_video.py:676)
        x = x.permute(0, 2, 1, 3, 4)
    # x: [B, F, 3, H, W]
    Fr = x.shape[1]
    assert Fr == cfg.frames, (Fr, cfg.frames)
    if cfg.tubelet == 1:
        fr = x.reshape(B * Fr, 3, cfg.img, cfg.img)
        emb = F.conv2d(fr, W["patch_w"].reshape(cfg.dim, 3, cfg.patch, cfg.patch), W.get("patch_b"), stride=cfg.patch)
        emb = emb.flatten(2).transpose(1, 2)  # [B*F, hp*hp, D]
    else:
        vol = x.permute(0, 2, 1, 3, 4)  # [B, 3, F, H, W]  (VivitTubeletEmbeddings)
        emb = F.conv3d(vol, W["patch_w"].reshape(cfg.dim, 3, cfg.tubelet, cfg.patch, cfg.patch), W.get("patch_b"),
                       stride=(cfg.tubelet, cfg.patch, cfg.patch))
        emb = emb.flatten(2).transpose(1, 2)  # [B, T'*hp*hp, D]
    if not cfg.joint_space_time and cfg.tubelet != 1:
        raise ValueError("tubelets imply joint space-time sequences")
    emb = emb + W["pos"][None]
    if cfg.prefix_tokens:
        pre = W["prefix"][None].expand(emb.shape[0], -1, -1)
        emb = torch.cat([pre, emb], dim=1)
    if cfg.pre_ln:  # LanguageBind pre_layrnorm (modeling_video.py:686)
        emb = F.layer_norm(emb, (cfg.dim,), W["pre_ln_w"], W["pre_ln_b"], cfg.ln_eps)
    return emb


def encoder_block(x: torch.Tensor, cfg: EncoderCfg, Lw: Dict[str, torch.Tensor], mx=False, first: bool = False) -> torch.Tensor:
    """(mx=True: every GEMM of the block -- temporal qkv / out-projection, qkv, attention out-projection, fc1, fc2 --
    through mx_linear: the MXFP8 mode's emulation. mx="folded": the same with qkv / fc1 (and, past the first block, LanguageBind's temporal qkv)
    in the LayerNorm-folded form the HIP path takes by default, mx_folded_ln_linear.) One pre-LN block. timm Block (x + ls1(attn(norm1 x)); x + ls2(mlp(norm2 x))); HF VivitLayer; LanguageBind
    CLIPEncoderLayer with its temporal sub-block first (modeling_video.py:133-179)."""
    D = cfg.dim
    lin = mx_linear if mx else F.linear
    if cfg.temporal_frames:
        t = cfg.temporal_frames
        bt, n, d = x.shape
        b = bt // t
        # time embed (:137-141): "(b t) n d -> (b n) t d", += temporal_embedding[:, :t]
        h = x.reshape(b, t, n, d).permute(0, 2, 1, 3).reshape(b * n, t, d)
        h = h + Lw["t_emb"][None, :t]
        x = h.reshape(b, n, t, d).permute(0, 2, 1, 3).reshape(bt, n, d)
        residual = x  # :144
        h = x.reshape(b, t, n, d).permute(0, 2, 1, 3).reshape(b * n, t, d)
        if mx == "folded" and not first:  # (the first block's temporal LayerNorm stays a kernel on the HIP path: its x comes from the embedding)
            t_in = lambda hh, w_, b_: mx_folded_ln_linear(hh, Lw["t_ln_w"], Lw["t_ln_b"], w_, b_, cfg.ln_eps)
            h = _mhsa(h, Lw["t_qkv_w"], Lw["t_qkv_b"], Lw["t_proj_w"], Lw["t_proj_b"], cfg.heads, lin, t_in)
        else:
            h = F.layer_norm(h, (D,), Lw["t_ln_w"], Lw["t_ln_b"], cfg.ln_eps)  # :147
            h = _mhsa(h, Lw["t_qkv_w"], Lw["t_qkv_b"], Lw["t_proj_w"], Lw["t_proj_b"], cfg.heads, lin)
        x = residual + h.reshape(b, n, t, d).permute(0, 2, 1, 3).reshape(bt, n, d)  # :155
    if mx == "folded":
        q_in = lambda hh, w_, b_: mx_folded_ln_linear(hh, Lw["ln1_w"], Lw["ln1_b"], w_, b_, cfg.ln_eps, D, (D // cfg.heads) ** -0.5 * 1.4426950408889634)
        h = _mhsa(x, Lw["qkv_w"], Lw["qkv_b"], Lw["proj_w"], Lw["proj_b"], cfg.heads, lin, q_in)
    else:
        h = F.layer_norm(x, (D,), Lw["ln1_w"], Lw["ln1_b"], cfg.ln_eps)
        h = _mhsa(h, Lw["qkv_w"], Lw["qkv_b"], Lw["proj_w"], Lw["proj_b"], cfg.heads, lin)
    if cfg.layerscale:
        h = h * Lw["ls1"]
    x = x + h
    if mx == "folded":
        h = lin(act_fn(cfg.act, mx_folded_ln_linear(x, Lw["ln2_w"], Lw["ln2_b"], Lw["fc1_w"], Lw["fc1_b"], cfg.ln_eps)), Lw["fc2_w"], Lw["fc2_b"])
    else:
        h = F.layer_norm(x, (D,), Lw["ln2_w"], Lw["ln2_b"], cfg.ln_eps)
        h = lin(act_fn(cfg.act, lin(h, Lw["fc1_w"], Lw["fc1_b"])), Lw["fc2_w"], Lw["fc2_b"])
    if cfg.layerscale:
        h = h * Lw["ls2"]
    return x + h


def encoder_forward(pix: torch.Tensor, cfg: EncoderCfg, W: Dict, mx=False) -> torch.Tensor:
    """VideoBackbone.forward -> [B, num_patches, D]:
    languagebind/__init__.py:79-103 (hidden_states[-2], 'noclass'), dinov2_video.py:132-154 (n={L-2}, prefix
    stripped, no final norm), vivit.py:100-118 (last layer + final LayerNorm, drop cls, (B,16,14,14,C)),
    siglip.py:142-151."""
    B = pix.shape[0]
    x = encoder_hidden(pix, cfg, W, mx)
    x = x[:, cfg.prefix_tokens:]
    return x.reshape(B, -1, cfg.dim)


def encoder_hidden(pix: torch.Tensor, cfg: EncoderCfg, W: Dict, mx=False) -> torch.Tensor:
    """Every token of every sequence after the last block run (and the final LayerNorm where the family applies one):
    [B * sequences, prefix + patches, D] -- hidden_states[-2] of the LanguageBind tower, timm's intermediate layer
    (patch and prefix tokens), VivitModel.last_hidden_state."""
    x = encoder_embed(pix, cfg, W)
    for li in range(cfg.layers):
        x = encoder_block(x, cfg, W["layers"][li], mx, first=li == 0)
    if cfg.final_ln:
        x = F.layer_norm(x, (cfg.dim,), W["final_ln_w"], W["final_ln_b"], cfg.ln_eps)
    return x


def map_pool(x: torch.Tensor, Pw: Dict[str, torch.Tensor], heads: int, act: str = "gelu_erf", eps: float = 1e-6) -> torch.Tensor:
    """The attention-pooling ("MAP") head timm's forward() ends with on the SigLIP towers (global_pool='map':
    AttentionPoolLatent, timm 0.9.10 -- ABSENT here, timm parity unpinned; the same head as transformers'
    SiglipMultiheadAttentionPoolingHead, on whose outputs this function is pinned, tests/golden/siglip_pool.npz): a learnt
    latent query attends over the final-norm tokens of a frame, out-projection, x + mlp(norm(x)), token 0.
    x [N, S, D] -> [N, D]. Pw: latent [1, D], q_w/q_b [D, D], kv_w/kv_b [2D, D], proj_w/b, norm_w/b, fc1_w/b, fc2_w/b."""
    N, S, D = x.shape
    hd = D // heads
    q = F.linear(Pw["latent"], Pw["q_w"], Pw["q_b"]).reshape(1, heads, 1, hd).expand(N, -1, -1, -1)
    kv = F.linear(x, Pw["kv_w"], Pw["kv_b"]).reshape(N, S, 2, heads, hd).permute(2, 0, 3, 1, 4)
    att = torch.softmax((q @ kv[0].transpose(-1, -2)) * hd**-0.5, dim=-1) @ kv[1]  # [N, heads, 1, hd]
    y = F.linear(att.transpose(1, 2).reshape(N, D), Pw["proj_w"], Pw["proj_b"])
    h = F.layer_norm(y, (D,), Pw["norm_w"], Pw["norm_b"], eps)
    return y + F.linear(act_fn(act, F.linear(h, Pw["fc1_w"], Pw["fc1_b"])), Pw["fc2_w"], Pw["fc2_b"])


def select_tokens(hidden: torch.Tensor, family: str, B: int, rule: Optional[str]) -> torch.Tensor:
    """The registry's token selections (merv/models/materialize.py:31-73) applied to encoder_hidden()'s tensor.
    languagebind (`token` kwarg, languagebind/__init__.py:88-101): None, "average", "classemb", "noclass", "classemb-at-first".
    dinov2 (substring of the id, dinov2_video.py:40-66,140-151): "all-tokens", "all-token-with-cls", "classemb-at-first",
      "cls" (the bare `dinov2-video`: timm forward = final norm + class token; hidden must come from the FULL depth with final_ln).
    vivit (vivit.py:106-118): "cls-token", "all-tokens", "all-no-cls", "all-no-cls-16frames", "classemb-at-first-16frames"."""
    D = hidden.shape[-1]
    if family == "languagebind":
        v = hidden.reshape(B, -1, hidden.shape[-2], D)  # [B, F, 257, D]
        if rule == "average":
            v = v.mean(-2)
        elif rule == "classemb":
            v = v[:, :, 0, :]
        elif rule == "noclass":
            v = v[:, :, 1:, :]
        elif rule == "classemb-at-first":
            cls = v[:, :, 0, :].mean(1, keepdim=True)
            v = torch.cat([cls, v[:, :, 1:, :].reshape(B, -1, D)], 1)
        return v.reshape(B, -1, D)
    if family == "dinov2":
        npre = 5
        patches, prefix = hidden[:, npre:], hidden[:, :npre]
        if rule == "cls":
            return hidden[:, 0].reshape(B, -1, D)
        if rule == "classemb-at-first":
            return torch.cat([prefix[:, :1].reshape(B, -1, D).mean(1, keepdim=True), patches.reshape(B, -1, D)], 1)
        if rule == "all-token-with-cls":
            return torch.cat([prefix[:, :1].reshape(B, -1, D), patches.reshape(B, -1, D)], 1)
        assert rule == "all-tokens", rule
        return patches.reshape(B, -1, D)
    if family == "vivit":
        if rule == "cls-token":
            return hidden[:, 0].unsqueeze(1)
        if rule == "all-no-cls-16frames":
            return hidden[:, 1:].reshape(B, 16 * 14 * 14, D)
        if rule == "all-no-cls":
            return hidden[:, 1:].reshape(B, 16, 14, 14, D)[:, ::2].reshape(B, 8 * 14 * 14, D)
        assert rule in ("all-tokens", "classemb-at-first-16frames"), rule  # neither branch of vivit.py:106-117 fires: unchanged
        return hidden
    raise ValueError(family)


# ------------------------------------------------------------------------------------------------------------
# a9. 3davg + linear projector -- merv/util/nn_utils.py:306-338, :22-32
# ------------------------------------------------------------------------------------------------------------
def adaptive_windows(S: int, O: int) -> List[Tuple[int, int]]:
    """torch AdaptiveAvgPool window rule: [floor(i*S/O), ceil((i+1)*S/O))."""
    return [((i * S) // O, -((-(i + 1) * S) // O)) for i in range(O)]


def projector_forward(tokens: torch.Tensor, T: int, S: int, out_size: int, proj_w: torch.Tensor,
                      proj_b: torch.Tensor) -> torch.Tensor:
    """AveragePooling3DProjector.forward (nn_utils.py:320-330) with output_frames == T (merv.py:158) and
    LinearProjector (:31-32). tokens [B, T*S*S, C] -> [B, T*out^2, llm]. The pool is restated with explicit
    windows (not by calling AdaptiveAvgPool3d) so that the window rule itself is under test."""
    B, N, C = tokens.shape
    assert N == T * S * S
    x = tokens.float().reshape(B, T, S, S, C)
    wins = adaptive_windows(S, out_size)
    rows = []
    for (y0, y1) in wins:
        cols = []
        for (x0, x1) in wins:
            cols.append(x[:, :, y0:y1, x0:x1].mean(dim=(2, 3)))
        rows.append(torch.stack(cols, dim=2))
    pooled = torch.stack(rows, dim=2)  # [B, T, O, O, C]
    pooled = pooled.reshape(B, T * out_size * out_size, C)
    return F.linear(pooled, proj_w, proj_b)


# ------------------------------------------------------------------------------------------------------------
# a10. cross-encoder fusion -- merv/util/nn_utils.py:455-521 (averagetoken=True, nn.MultiheadAttention heads=1)
# ------------------------------------------------------------------------------------------------------------
def fusion_forward(V: Sequence[torch.Tensor], Fw: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """V: E tensors [B, T, C]. Fw: reference state-dict entries of CrossAttentionAdapterLearnableQuery:
    Q [1,Ed], attention.q_proj_weight [Ed,Ed], attention.k_proj_weight [Ed,C], attention.in_proj_bias [3Ed].
    v_proj / out_proj only feed the discarded MHA output `p` (nn_utils.py:512) and are not needed.
    Returns (sum_e w_e V_e  [B,T,C], w [B,E])."""
    Vs = torch.stack([v.float() for v in V], dim=1)  # [B, E, T, C]   (:500)
    B, E, T, C = Vs.shape
    Ed = Fw["Q"].shape[1]
    Vbar = Vs.mean(2)  # :506
    bias = Fw["attention.in_proj_bias"]
    q = F.linear(Fw["Q"], Fw["attention.q_proj_weight"], bias[:Ed])  # [1, Ed]
    k = F.linear(Vbar, Fw["attention.k_proj_weight"], bias[Ed : 2 * Ed])  # [B, E, Ed]
    scores = (k @ q.t()).squeeze(-1) / math.sqrt(Ed)  # single head: head_dim == Ed
    w = scores.softmax(dim=-1)  # [B, E]
    out = torch.einsum("be,betc->btc", w, Vs)  # torch.bmm(weights, V) (:521)
    return out, w


def fusion_fold_u(Fw: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Host-side fold used by the HIP path's binding, restated here so tests can check it independently:
    u = Wk^T (Wq Q + bq) / sqrt(Ed). (bk.q is common to all encoders and cancels in the softmax.)"""
    Ed = Fw["Q"].shape[1]
    bias = Fw["attention.in_proj_bias"].double()
    q = Fw["attention.q_proj_weight"].double() @ Fw["Q"].double()[0] + bias[:Ed]
    return (Fw["attention.k_proj_weight"].double().t() @ q / math.sqrt(Ed)).float()


# ------------------------------------------------------------------------------------------------------------
# a11. splice -- merv/models/vidlms/merv.py:633-664
# ------------------------------------------------------------------------------------------------------------
def splice(input_embeddings: torch.Tensor, fused: torch.Tensor, bos_token_length: int = 1,
           attention_mask: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
           ignore_index: int = -100):
    """cat[emb[:, :bos], fused, emb[:, bos:]] (+ attention mask of True, labels of IGNORE_INDEX over the visual span)."""
    b = bos_token_length
    emb = torch.cat([input_embeddings[:, :b], fused, input_embeddings[:, b:]], dim=1)
    am = lab = None
    if attention_mask is not None:
        vis = torch.full(fused.shape[:2], True, dtype=attention_mask.dtype)
        am = torch.cat([attention_mask[:, :b], vis, attention_mask[:, b:]], dim=1)
    if labels is not None:
        vis = torch.full(fused.shape[:2], ignore_index, dtype=labels.dtype)
        lab = torch.cat([labels[:, :b], vis, labels[:, b:]], dim=1)
    return emb, am, lab


# ------------------------------------------------------------------------------------------------------------
# f-4. training-mode batch assembly and loss -- merv/models/vidlms/merv.py:612-734, restated line by line and pinned (since
# round 3) on the reference's own MERV.forward body run from its AST (tools/make_goldens.py, tests/golden/merv_forward.npz,
# tests/test_oracle_goldens.py). Gradients of the projector / fusion parameters are torch autograd through
# projector_forward / fusion_forward above, exactly what the reference's loss.backward() differentiates.
# ------------------------------------------------------------------------------------------------------------
def assemble_training_batch(input_embeddings: torch.Tensor, fused: torch.Tensor, attention_mask: torch.Tensor,
                            labels: torch.Tensor, multimodal_indices: torch.Tensor, bos_token_length: int = 1,
                            ignore_index: int = -100):
    """Multimodal rows: visual span after BOS, mask True, labels IGNORE (:633-664). Unimodal rows: padded at the end by
    `padcount` spans of zeros / False / IGNORE (:676-713), stacked under the multimodal rows (:716-719)."""
    mm = multimodal_indices
    emb_mm, am_mm, lab_mm = splice(input_embeddings[mm], fused, bos_token_length, attention_mask[mm], labels[mm], ignore_index)
    uni = torch.tensor([i for i in range(input_embeddings.shape[0]) if i not in set(mm.tolist())], dtype=torch.long)
    if len(uni) == 0:
        return emb_mm, am_mm, lab_mm
    Tv = fused.shape[1]
    padcount = (emb_mm.shape[1] - input_embeddings.shape[1]) // Tv  # == 1
    emb_u = torch.cat([input_embeddings[uni]] + [torch.zeros(len(uni), Tv, input_embeddings.shape[2])] * padcount, dim=1)
    am_u = torch.cat([attention_mask[uni]] + [torch.full((len(uni), Tv), False, dtype=attention_mask.dtype)] * padcount, dim=1)
    lab_u = torch.cat([labels[uni]] + [torch.full((len(uni), Tv), ignore_index, dtype=labels.dtype)] * padcount, dim=1)
    return torch.vstack([emb_mm, emb_u]), torch.vstack([am_mm, am_u]), torch.vstack([lab_mm, lab_u])


def causal_lm_loss(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """The HF causal-LM loss the reference returns (llm_backbone(..., labels=...), merv.py:723-734): token t predicts
    label t+1, mean cross-entropy over the labels that are not IGNORE_INDEX."""
    shift_logits = logits[:, :-1].reshape(-1, logits.shape[-1]).float()
    shift_labels = labels[:, 1:].reshape(-1)
    return F.cross_entropy(shift_logits, shift_labels, ignore_index=ignore_index)


def cosine_schedule_with_warmup(step: int, num_warmup_steps: int, num_training_steps: int) -> float:
    """LR multiplier of transformers.get_cosine_schedule_with_warmup (num_cycles = 0.5), the scheduler the reference
    builds (training/strategies/fsdp.py:291)."""
    if step < num_warmup_steps:
        return step / max(1, num_warmup_steps)
    progress = (step - num_warmup_steps) / max(1, num_training_steps - num_warmup_steps)
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress)))


# ------------------------------------------------------------------------------------------------------------
# whole path: a4-a10 (merv.py:562-609)
# ------------------------------------------------------------------------------------------------------------
def visual_path_forward(pixels: Sequence[torch.Tensor], cfgs: Sequence[EncoderCfg], enc_W: Sequence[Dict],
                        proj_W: Sequence[Tuple[torch.Tensor, torch.Tensor]], Fw: Dict[str, torch.Tensor],
                        out_size: int = 8):
    """Encoders sequentially (merv.py:563-566) -> reshape [B,T,S,C] (:576-585) -> projectors (:587-589) ->
    fusion (:607-609). Returns (fused [B, T*out^2, llm], weights [B,E], projected list)."""
    projected = []
    for pix, cfg, W, (pw, pb) in zip(pixels, cfgs, enc_W, proj_W):
        tok = encoder_forward(pix, cfg, W)
        projected.append(projector_forward(tok, cfg.t_out, cfg.hp, out_size, pw, pb))
    fused, w = fusion_forward(projected, Fw)
    return fused, w, projected


# ------------------------------------------------------------------------------------------------------------
# seeded random weights in canonical layout (synthetic data for tests / bench; there are no checkpoints here)
# ------------------------------------------------------------------------------------------------------------
def random_encoder_weights(cfg: EncoderCfg, seed: int) -> Dict:
    g = torch.Generator().manual_seed(seed)
    D, Mh = cfg.dim, cfg.mlp_dim

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    k = 3 * cfg.tubelet * cfg.patch * cfg.patch
    P = cfg.s_out * (cfg.t_out if cfg.joint_space_time else 1)
    W = {
        "patch_w": rn(D, k, std=k**-0.5),
        "pos": rn(P, D),
        "layers": [],
    }
    if cfg.name != "languagebind":  # CLIP conv has no bias (HF CLIPVisionEmbeddings)
        W["patch_b"] = rn(D)
    if cfg.prefix_tokens:
        W["prefix"] = rn(cfg.prefix_tokens, D)
    if cfg.pre_ln:
        W["pre_ln_w"] = 1.0 + rn(D, std=0.1)
        W["pre_ln_b"] = rn(D, std=0.1)
    if cfg.final_ln:
        W["final_ln_w"] = 1.0 + rn(D, std=0.1)
        W["final_ln_b"] = rn(D, std=0.1)
    for _ in range(cfg.layers):
        Lw = {
            "ln1_w": 1.0 + rn(D, std=0.1), "ln1_b": rn(D, std=0.1),
            "qkv_w": rn(3 * D, D, std=D**-0.5), "qkv_b": rn(3 * D),
            "proj_w": rn(D, D, std=D**-0.5), "proj_b": rn(D),
            "ln2_w": 1.0 + rn(D, std=0.1), "ln2_b": rn(D, std=0.1),
            "fc1_w": rn(Mh, D, std=D**-0.5), "fc1_b": rn(Mh),
            "fc2_w": rn(D, Mh, std=Mh**-0.5), "fc2_b": rn(D),
        }
        if cfg.layerscale:
            Lw["ls1"] = 0.5 + rn(D, std=0.2)
            Lw["ls2"] = 0.5 + rn(D, std=0.2)
        if cfg.temporal_frames:
            Lw.update({
                "t_emb": rn(cfg.temporal_frames, D, std=D**-0.5),
                "t_ln_w": 1.0 + rn(D, std=0.1), "t_ln_b": rn(D, std=0.1),
                "t_qkv_w": rn(3 * D, D, std=D**-0.5), "t_qkv_b": rn(3 * D),
                "t_proj_w": rn(D, D, std=D**-0.5), "t_proj_b": rn(D),
            })
        W["layers"].append(Lw)
    return W


def random_projector_weights(C: int, llm: int, seed: int):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(llm, C, generator=g) * C**-0.5, torch.randn(llm, generator=g) * 0.02


def random_fusion_weights(llm: int, embed_dim: int, seed: int) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    return {
        "Q": torch.randn(1, embed_dim, generator=g) * 0.5,
        "attention.q_proj_weight": torch.randn(embed_dim, embed_dim, generator=g) * embed_dim**-0.5,
        "attention.k_proj_weight": torch.randn(embed_dim, llm, generator=g) * llm**-0.5,
        "attention.v_proj_weight": torch.randn(embed_dim, llm, generator=g) * llm**-0.5,
        "attention.in_proj_bias": torch.randn(3 * embed_dim, generator=g) * 0.02,
    }
