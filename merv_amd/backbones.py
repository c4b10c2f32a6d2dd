"""
Video backbones with the reference's class names, constructor arguments, properties and registry keys
(merv/models/backbones/video/*.py, merv/models/materialize.py:31-73,107-129) whose forward() runs the HIP encoder.

Differences from the reference, all forced by this environment (no network, no timm):
  * weights are not downloaded in __init__; pass `weights=` (an upstream state dict of the matching family or a
    canonical dict, see merv_amd/weights.py) or `weights="random"` for seeded synthetic parameters;
  * every token selection of the LanguageBind / DINOv2 / ViViT registry keys is wired (a slice, a concatenation or a mean --
    merv_mean_rows -- of the encoder's full token tensor, merv_encoder_forward_select); the B/16-224 SigLIP keys without
    `all-no-cls` return timm's attention-pooled (MAP head) feature per frame (siglip.py:46-63 leaves the timm forward in place),
    built here from the library's GEMM / LayerNorm kernels and merv_map_pool_attention; `siglip-...-classemb-at-first` fails
    inside the reference itself (a tuple is reshaped, siglip.py:148-149) and raises NotImplementedError when constructed;
  * `video_transform` is the GPU implementation of the reference's CPU PIL / torchvision pipelines
    (merv_amd/preprocess.py: Pillow-bit-exact resize + ToTensor + Normalize; LanguageBind's torch pipeline with the
    random flip made an explicit, default-off switch): it takes load_video()'s uint8 [F,3,H,W] tensor on the device.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import weights as W
from .encoder import EncoderSpec, HipEncoder


class VideoBackbone(nn.Module):
    """base_video.py:56-105"""

    def __init__(self, video_backbone_id: str, image_resize_strategy: str, default_image_size: int = 224,
                 num_frames: int = 8) -> None:
        super().__init__()
        self.identifier = video_backbone_id
        self.image_resize_strategy = image_resize_strategy
        self.default_image_size = default_image_size
        self.num_frames = num_frames
        self.featurizer: Optional[HipEncoder] = None
        self.video_transform = None

    def get_video_transform(self):
        return self.video_transform

    def get_fsdp_wrapping_policy(self) -> Callable:
        raise NotImplementedError("encoders are frozen and forward-only on the HIP path (merv.py:315-384)")

    # -- shared plumbing -------------------------------------------------------------------------------------
    def _build(self, spec: EncoderSpec, weights, device, ingest: Callable, ln_fold: bool = True) -> None:
        self.spec = spec
        if weights is None:
            raise ValueError(f"{type(self).__name__}: pass weights=<state dict> or weights='random' (no hub access here)")
        if isinstance(weights, str) and weights == "random":
            canon = random_weights(spec, seed=spec.dim + spec.frames)
        elif "layers" in weights and "patch_w" in weights:
            canon = weights
        else:
            canon = ingest(weights)
        self.featurizer = HipEncoder(spec, canon, device, ln_fold=ln_fold)
        # the reference's per-encoder CPU transform (row a3), as HIP kernels: uint8 [F,3,H,W] -> this encoder's layout
        from .preprocess import transform_for
        self.video_transform = transform_for(spec.name, torch.float32)

    def forward(self, video_values: torch.Tensor, is_image: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.featurizer.forward(video_values)

    @property
    def selects_spec_patches(self) -> bool:
        """True when forward() is exactly the encoder's patch-token output (`featurizer.forward(video)`): the only selection
        MervVisualPath fuses -- it drives the featurizers directly and sizes the projector from spec.t_out x spec.s_out.
        Every other registry selection (class token, averages, class-token-first, pooled heads) is served by forward() alone."""
        return True

    def _mean_rows(self, x: torch.Tensor) -> torch.Tensor:
        """[..., R, D] bf16 -> [..., D]: torch.mean(dim=-2) of a bf16 tensor (fp32 accumulation, one rounding), as a HIP kernel."""
        from ._lib import check, ptr
        x = x.contiguous()
        R, D = x.shape[-2], x.shape[-1]
        out = torch.empty(*x.shape[:-2], D, dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device(x.device):
            check(self.featurizer._lib.merv_mean_rows(ptr(x), ptr(out), x.numel() // (R * D), R, D, R,
                                                      torch.cuda.current_stream(x.device).cuda_stream), "merv_mean_rows")
        return out

    @property
    def embed_dim(self) -> int:
        return self.spec.dim

    @property
    def num_patches(self) -> int:
        return self.spec.num_patches

    @property
    def spatial_resolution(self) -> int:
        return self.spec.s_out

    @property
    def temporal_resolution(self) -> int:
        assert self.num_patches % self.spatial_resolution == 0
        return self.num_patches // self.spatial_resolution

    @property
    def half_precision_dtype(self) -> torch.dtype:
        return torch.bfloat16


_BF16_STORED = ("patch_w", "pos", "prefix", "qkv_w", "proj_w", "fc1_w", "fc2_w", "t_qkv_w", "t_proj_w")


def random_weights(spec: EncoderSpec, seed: int, device="cpu", bf16_exact: bool = False) -> Dict:
    """Seeded synthetic parameters in canonical layout (fp32), generated on `device` (there are no checkpoints here).
    `bf16_exact=True` rounds the tensors the HIP path stores in bf16 (GEMM weights, position / prefix rows) to
    bf16-representable values, as the reference's `vidlm.to(torch.bfloat16)` (scripts/quick_start.py:12) does to its
    parameters: a checker that is handed the same dict then sees exactly the values the kernels multiply."""
    g = torch.Generator(device=device).manual_seed(seed)
    D, Mh = spec.dim, spec.mlp_dim

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g, device=device) * std

    P = spec.s_out * (spec.t_out if spec.joint_space_time else 1)
    out = {"patch_w": rn(D, spec.k_true, std=spec.k_true**-0.5), "pos": rn(P, D), "layers": []}
    if spec.name != "languagebind":
        out["patch_b"] = rn(D)
    if spec.prefix_tokens:
        out["prefix"] = rn(spec.prefix_tokens, D)
    if spec.pre_ln:
        out["pre_ln_w"], out["pre_ln_b"] = 1 + rn(D, std=0.1), rn(D, std=0.1)
    if spec.final_ln:
        out["final_ln_w"], out["final_ln_b"] = 1 + rn(D, std=0.1), rn(D, std=0.1)
    for _ in range(spec.layers):
        Lw = {"ln1_w": 1 + rn(D, std=0.1), "ln1_b": rn(D, std=0.1), "qkv_w": rn(3 * D, D, std=D**-0.5), "qkv_b": rn(3 * D),
              "proj_w": rn(D, D, std=D**-0.5), "proj_b": rn(D), "ln2_w": 1 + rn(D, std=0.1), "ln2_b": rn(D, std=0.1),
              "fc1_w": rn(Mh, D, std=D**-0.5), "fc1_b": rn(Mh), "fc2_w": rn(D, Mh, std=Mh**-0.5), "fc2_b": rn(D)}
        if spec.layerscale:
            Lw["ls1"], Lw["ls2"] = 0.5 + rn(D, std=0.2), 0.5 + rn(D, std=0.2)
        if spec.temporal_frames:
            Lw.update({"t_emb": rn(spec.temporal_frames, D, std=D**-0.5), "t_ln_w": 1 + rn(D, std=0.1),
                       "t_ln_b": rn(D, std=0.1), "t_qkv_w": rn(3 * D, D, std=D**-0.5), "t_qkv_b": rn(3 * D),
                       "t_proj_w": rn(D, D, std=D**-0.5), "t_proj_b": rn(D)})
        out["layers"].append(Lw)
    if bf16_exact:
        for d in [out] + out["layers"]:
            for k in _BF16_STORED:
                if k in d:
                    d[k] = d[k].to(torch.bfloat16).float()
    return out


def random_map_pool_weights(D: int, mlp: int, seed: int) -> Dict:
    """Seeded synthetic parameters of the attention-pooling head (timm AttentionPoolLatent layout, see weights.from_timm_attn_pool)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *shape, std=0.02: torch.randn(*shape, generator=g) * std
    return {"latent": rn(1, D, std=0.5), "q_w": rn(D, D, std=D**-0.5), "q_b": rn(D), "kv_w": rn(2 * D, D, std=D**-0.5), "kv_b": rn(2 * D),
            "proj_w": rn(D, D, std=D**-0.5), "proj_b": rn(D), "norm_w": 1 + rn(D, std=0.1), "norm_b": rn(D, std=0.1),
            "fc1_w": rn(mlp, D, std=D**-0.5), "fc1_b": rn(mlp), "fc2_w": rn(D, mlp, std=mlp**-0.5), "fc2_b": rn(D)}


def weights_to(W: Dict, device) -> Dict:
    """Copy of a canonical weight dict on `device` (e.g. to hand the parameters of a GPU-resident path to a host checker)."""
    out = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in W.items() if k != "layers"}
    out["layers"] = [{k: v.to(device) for k, v in L.items()} for L in W["layers"]]
    return out


class LangBindVideoBackbone(VideoBackbone):
    """languagebind/__init__.py:33-135 -- CLIP ViT-L/14 + temporal attention, hidden_states[-2]."""

    def __init__(self, video_backbone_id: str, image_resize_strategy: str, default_image_size: int = 224,
                 num_frames: int = 8, token: Optional[str] = None, weights=None, device="cuda:0",
                 hidden_act: str = "gelu_erf", layers: int = 23, ln_fold: bool = True) -> None:
        super().__init__(video_backbone_id, image_resize_strategy, default_image_size, num_frames)
        assert "languagebind-video" in video_backbone_id, video_backbone_id
        assert image_resize_strategy == "resize-naive"  # languagebind/__init__.py:64
        if token not in (None, "average", "classemb", "noclass", "classemb-at-first"):
            raise ValueError(f"LanguageBind token selection `{token}` does not exist (languagebind/__init__.py:88-98)")
        self.token = token
        spec = EncoderSpec("languagebind", 1024, 16, 4096, layers, 14, 1, default_image_size, num_frames, "BCFHW", 1, False,
                           True, False, False, 8, hidden_act, 1e-5)
        self._build(spec, weights, device, lambda sd: W.from_languagebind_vision(sd, n_layers=layers), ln_fold)

    def forward(self, video_values: torch.Tensor, is_image: Optional[torch.Tensor] = None) -> torch.Tensor:
        """languagebind/__init__.py:79-103: hidden_states[-2] [B, F, 257, D], then the `token` rule."""
        if self.token == "noclass":
            return self.featurizer.forward(video_values)
        B, D = video_values.shape[0], self.spec.dim
        v = self.featurizer.forward(video_values, select="all").view(B, -1, 257, D)
        if self.token == "average":
            return self._mean_rows(v)
        if self.token == "classemb":
            return v[:, :, 0, :].contiguous()
        if self.token == "classemb-at-first":
            cls = self._mean_rows(v[:, :, 0, :]).unsqueeze(1)  # mean over the frames of the per-frame class tokens
            return torch.cat([cls, v[:, :, 1:, :].reshape(B, -1, D)], 1)
        return v.reshape(B, -1, D)

    @property
    def selects_spec_patches(self) -> bool:
        return self.token == "noclass"

    @property
    def num_patches(self) -> int:  # languagebind/__init__.py:113-126 (classemb-at-first: the class token is not counted)
        return self.num_frames * {None: 257, "average": 1, "classemb": 1, "noclass": 256, "classemb-at-first": 256}[self.token]

    @property
    def spatial_resolution(self) -> int:
        return self.num_patches // self.num_frames

    @property
    def default_video_resolution(self) -> Tuple[int, int, int, int]:
        return (3, self.num_frames, 224, 224)


class DinoV2VideoBackbone(VideoBackbone):
    """dinov2_video.py:27-179 -- timm vit_large_patch14_reg4_dinov2, get_intermediate_layers(n={L-2})."""

    def __init__(self, video_backbone_id: str, image_resize_strategy: str, default_image_size: int = 224,
                 num_frames: int = 8, weights=None, device="cuda:0", layers: Optional[int] = None, ln_fold: bool = True) -> None:
        super().__init__(video_backbone_id, image_resize_strategy, default_image_size, num_frames)
        # dinov2_video.py:46-66: ids with "all-token" / "classemb-at-first" read block L-2 (no final norm); the bare id keeps timm's
        # forward(): all L blocks, the final norm, the class token of every frame
        self.pooled = not ("all-token" in video_backbone_id or "classemb-at-first" in video_backbone_id)
        if layers is None:
            layers = 24 if self.pooled else 23
        spec = EncoderSpec("dinov2", 1024, 16, 4096, layers, 14, 1, default_image_size, num_frames, "BFCHW", 5, False, False,
                           self.pooled, True, 0, "gelu_erf", 1e-6)
        self._build(spec, weights, device,
                    lambda sd: W.from_timm_vit(sd, n_layers=layers, grid=default_image_size // 14), ln_fold)

    @property
    def default_video_resolution(self) -> Tuple[int, int, int, int]:
        return (self.num_frames, 3, self.default_image_size, self.default_image_size)


    def forward(self, video_values: torch.Tensor, is_image: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dinov2_video.py:132-154."""
        ident = self.identifier
        if "all-tokens" in ident:
            return self.featurizer.forward(video_values)
        B, D = video_values.shape[0], self.spec.dim
        v = self.featurizer.forward(video_values, select="all").view(B, -1, 5 + self.spec.s_out, D)  # [B, F, 5 + 256, D]
        if self.pooled:
            return v[:, :, 0, :].contiguous()
        patches = v[:, :, 5:, :].reshape(B, -1, D)
        cls = v[:, :, 0, :]  # prefix[:, :1]: the class token; the four register tokens are dropped
        if "classemb-at-first" in ident:
            cls = self._mean_rows(cls).unsqueeze(1)
        return torch.cat([cls, patches], 1)

    @property
    def selects_spec_patches(self) -> bool:
        return "all-tokens" in self.identifier

    @property
    def num_patches(self) -> int:  # dinov2_video.py:164-170, quirks kept ("all-token-with-cls" does not contain "all-tokens")
        if "classemb-at-first" in self.identifier or "all-tokens" in self.identifier:
            return self.num_frames * self.spec.s_out
        return self.num_frames

    @property
    def spatial_resolution(self) -> int:
        return self.num_patches // self.num_frames


class ViVITVideoBackbone(VideoBackbone):
    """vivit.py:24-155 -- HF VivitModel (google/vivit-b-16x2-kinetics400), last_hidden_state[:, 1:]."""

    def __init__(self, video_backbone_id: str, image_resize_strategy: str, default_image_size: int = 224,
                 num_frames: int = 32, weights=None, device="cuda:0", layers: int = 12, ln_fold: bool = True) -> None:
        super().__init__(video_backbone_id, image_resize_strategy, default_image_size, num_frames)
        self.video_backbone_id = video_backbone_id
        spec = EncoderSpec("vivit", 768, 12, 3072, layers, 16, 2, default_image_size, num_frames, "BFCHW", 1, True, False, True,
                           False, 0, "gelu_tanh", 1e-6)
        self._build(spec, weights, device, lambda sd: W.from_hf_vivit(sd, n_layers=layers), ln_fold)

    @property
    def default_video_resolution(self) -> Tuple[int, int, int, int]:
        return (self.num_frames, 3, self.default_image_size, self.default_image_size)


    def forward(self, video_values: torch.Tensor, is_image: Optional[torch.Tensor] = None) -> torch.Tensor:
        """vivit.py:100-118 on last_hidden_state [B, 3137, D]."""
        ident = self.video_backbone_id
        if "all-no-cls-16frames" in ident:
            return self.featurizer.forward(video_values)
        v = self.featurizer.forward(video_values, select="all")
        B, D = v.shape[0], v.shape[-1]
        if "cls-token" in ident:
            return v[:, 0].unsqueeze(1).contiguous()
        if "all-no-cls" in ident:  # every second of the 16 tubelet slots
            return v[:, 1:].reshape(B, 16, 14, 14, D)[:, ::2].reshape(B, 8 * 14 * 14, D)
        return v  # all-tokens, classemb-at-first-16frames: neither branch of vivit.py:106-117 fires

    @property
    def selects_spec_patches(self) -> bool:
        return "all-no-cls-16frames" in self.video_backbone_id

    @property
    def num_patches(self) -> int:  # vivit.py:128-142
        ident = self.video_backbone_id
        if "cls-token" in ident:
            return 1
        if "all-tokens" in ident:
            return 3137
        if "all-no-cls-16frames" in ident or "classemb-at-first" in ident:
            return 3136
        if "all-no-cls" in ident:
            return 3136 // 2
        raise NotImplementedError(ident)

    @property
    def spatial_resolution(self) -> int:  # vivit.py:145-151
        if "all-no-cls" in self.video_backbone_id or "classemb-at-first" in self.video_backbone_id:
            return 196
        return self.num_patches


class SiglipVideoBackbone(VideoBackbone):
    """siglip.py:35-174 -- timm vit_base_patch16_siglip_224. `*-all-no-cls`: get_intermediate_layers(n={L-2}), every patch token.
    Every other id keeps timm's forward() (siglip.py:46-63 only patches it for all-no-cls / classemb-at-first): all L blocks, the
    final norm and the attention-pooling head (global_pool='map'), one feature per frame."""

    def __init__(self, video_backbone_id: str, image_resize_strategy: str, default_image_size: int = 224,
                 num_frames: int = 8, weights=None, device="cuda:0", layers: Optional[int] = None, ln_fold: bool = True,
                 pool_weights=None) -> None:
        super().__init__(video_backbone_id, image_resize_strategy, default_image_size, num_frames)
        if not video_backbone_id.startswith("siglip-vit-b16-224px"):
            raise NotImplementedError(f"`{video_backbone_id}`: only the B/16 224 px tower is wired")
        if "classemb-at-first" in video_backbone_id:
            raise NotImplementedError(f"`{video_backbone_id}` fails inside the reference itself: get_intermediate_layers(return_prefix_tokens="
                                      "True) hands forward() a tuple, which it reshapes (siglip.py:56-63,148-149)")
        self.class_token = "all-no-cls" not in video_backbone_id  # siglip.py:46-49 (its name for "pooled output")
        if layers is None:
            layers = 12 if self.class_token else 11
        spec = EncoderSpec("siglip", 768, 12, 3072, layers, 16, 1, default_image_size, num_frames, "BFCHW", 0, False, False,
                           self.class_token, False, 0, "gelu_erf", 1e-6)
        self._build(spec, weights, device, lambda sd: W.from_timm_vit(sd, n_layers=layers), ln_fold)
        self.pool_weights = None
        if self.class_token:
            if pool_weights is None and isinstance(weights, dict) and any(k.endswith("attn_pool.latent") for k in weights):
                pool_weights = weights  # an upstream timm state dict carries attn_pool.* beside the blocks
            if isinstance(pool_weights, dict) and "latent" in pool_weights:
                pw = pool_weights
            elif isinstance(pool_weights, dict):
                pw = W.from_timm_attn_pool(pool_weights)
            elif isinstance(weights, str) and weights == "random":
                pw = random_map_pool_weights(spec.dim, spec.mlp_dim, seed=spec.dim + 7)
            else:
                raise ValueError(f"`{video_backbone_id}` returns the attention-pooled feature: pass pool_weights=<timm attn_pool.* state "
                                 "dict or MAP-head dict> (nothing is silently random)")
            self.pool_weights = {k: v.detach().float().cpu() for k, v in pw.items()}  # canonical fp32 copy (inspection / checks)
            dev = torch.device(device)
            bf = lambda k: self.pool_weights[k].to(dev, torch.bfloat16).contiguous()
            f32 = lambda k: self.pool_weights[k].to(dev, torch.float32).contiguous()
            self._pool = {"kv_w": bf("kv_w"), "kv_b": f32("kv_b"), "proj_w": bf("proj_w"), "proj_b": f32("proj_b"),
                          "norm_w": f32("norm_w"), "norm_b": f32("norm_b"), "fc1_w": bf("fc1_w"), "fc1_b": f32("fc1_b"),
                          "fc2_w": bf("fc2_w"), "fc2_b": f32("fc2_b"),
                          # the query does not depend on the input: q Linear of the latent, once
                          "q": torch.nn.functional.linear(self.pool_weights["latent"], self.pool_weights["q_w"],
                                                          self.pool_weights["q_b"]).reshape(-1).to(dev, torch.float32).contiguous()}

    def forward(self, video_values: torch.Tensor, is_image: Optional[torch.Tensor] = None) -> torch.Tensor:
        """siglip.py:142-151."""
        if not self.class_token:
            return self.featurizer.forward(video_values)
        from . import ops
        from ._lib import check, ptr
        B, D, S = video_values.shape[0], self.spec.dim, self.spec.s_out
        x = self.featurizer.forward(video_values, select="all").view(-1, D)  # [B*F*196, D]: blocks + final norm
        nseq = x.shape[0] // S
        P = self._pool
        with torch.cuda.device(x.device):
            kv = ops.gemm(x, P["kv_w"], P["kv_b"])  # [B*F*196, 2D]
            att = torch.empty(nseq, D, dtype=torch.bfloat16, device=x.device)
            check(self.featurizer._lib.merv_map_pool_attention(ptr(kv), ptr(P["q"]), ptr(att), nseq, S, self.spec.heads, 64**-0.5,
                                                               torch.cuda.current_stream(x.device).cuda_stream), "merv_map_pool_attention")
            y = ops.gemm(att, P["proj_w"], P["proj_b"])
            h = ops.gemm(ops.layernorm(y, P["norm_w"], P["norm_b"], 1e-6), P["fc1_w"], P["fc1_b"], act="gelu_erf")
            out = ops.gemm(h, P["fc2_w"], P["fc2_b"], res=y)  # x + mlp(norm(x)); latent_len = 1: token 0 is the row
        return out.view(B, -1, D)

    @property
    def selects_spec_patches(self) -> bool:
        return not self.class_token

    @property
    def num_patches(self) -> int:  # siglip.py:161-166
        return self.num_frames if self.class_token else self.num_frames * self.spec.s_out

    @property
    def spatial_resolution(self) -> int:
        return self.num_patches // self.num_frames

    @property
    def default_video_resolution(self) -> Tuple[int, int, int, int]:
        return (self.num_frames, 3, self.default_image_size, self.default_image_size)


# === Video Backbone Registry (merv/models/materialize.py:31-73) -- same keys, same kwargs ===
VIDEO_BACKBONES = {
    "dinov2-video": {"cls": DinoV2VideoBackbone, "kwargs": {"default_image_size": 224}},
    "dinov2-video-all-tokens": {"cls": DinoV2VideoBackbone, "kwargs": {"default_image_size": 224}},
    "dinov2-video-all-token-with-cls": {"cls": DinoV2VideoBackbone, "kwargs": {"default_image_size": 224}},
    "dinov2-video-classemb-at-first": {"cls": DinoV2VideoBackbone, "kwargs": {"default_image_size": 224}},
    "languagebind-video": {"cls": LangBindVideoBackbone, "kwargs": {"default_image_size": 224}},
    "languagebind-video-averagetoken": {"cls": LangBindVideoBackbone, "kwargs": {"default_image_size": 224, "token": "average"}},
    "languagebind-video-classemb": {"cls": LangBindVideoBackbone, "kwargs": {"default_image_size": 224, "token": "classemb"}},
    "languagebind-video-noclass": {"cls": LangBindVideoBackbone, "kwargs": {"default_image_size": 224, "token": "noclass"}},
    "languagebind-video-classemb-at-first": {"cls": LangBindVideoBackbone,
                                             "kwargs": {"default_image_size": 224, "token": "classemb-at-first"}},
    "vivit-google-b-cls-token": {"cls": ViVITVideoBackbone, "kwargs": {"default_image_size": 224}},
    "vivit-google-b-all-tokens": {"cls": ViVITVideoBackbone, "kwargs": {"default_image_size": 224}},
    "vivit-google-b-all-no-cls": {"cls": ViVITVideoBackbone, "kwargs": {"default_image_size": 224}},
    "vivit-google-b-all-no-cls-16frames": {"cls": ViVITVideoBackbone, "kwargs": {"default_image_size": 224}},
    "vivit-google-b-classemb-at-first-16frames": {"cls": ViVITVideoBackbone, "kwargs": {"default_image_size": 224}},
    "siglip-vit-b16-224px": {"cls": SiglipVideoBackbone, "kwargs": {"default_image_size": 224}},
    "siglip-vit-b16-224px-all-tokens": {"cls": SiglipVideoBackbone, "kwargs": {"default_image_size": 224}},
    "siglip-vit-b16-224px-all-no-cls": {"cls": SiglipVideoBackbone, "kwargs": {"default_image_size": 224}},
    "siglip-vit-b16-224px-classemb-at-first": {"cls": SiglipVideoBackbone, "kwargs": {"default_image_size": 224}},
}


def get_video_backbone_and_transform(video_backbone_ids: List[str], image_resize_strategy: str, num_frames: List[int],
                                     weights: Optional[List] = None, device="cuda:0"):
    """materialize.py:107-129. `weights[i]`: state dict / canonical dict / "random" for backbone i."""
    video_backbones, video_transforms = [], []
    for i, (video_backbone_id, num_frame) in enumerate(zip(video_backbone_ids, num_frames)):
        if video_backbone_id in VIDEO_BACKBONES:
            cfg = VIDEO_BACKBONES[video_backbone_id]
            bb = cfg["cls"](video_backbone_id, image_resize_strategy, num_frames=num_frame,
                            weights=None if weights is None else weights[i], device=device, **cfg["kwargs"])
            video_backbones.append(bb)
            video_transforms.append(bb.get_video_transform())
        else:
            raise ValueError(f"Video Backbone `{video_backbone_id}` is not supported!")
    return video_backbones, video_transforms
