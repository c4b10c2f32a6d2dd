"""
Video backbones with the reference's class names, constructor arguments, properties and registry keys
(merv/models/backbones/video/*.py, merv/models/materialize.py:31-73,107-129) whose forward() runs the HIP encoder.

Differences from the reference, all forced by this environment (no network, no timm):
  * weights are not downloaded in __init__; pass `weights=` (an upstream state dict of the matching family or a
    canonical dict, see merv_amd/weights.py) or `weights="random"` for seeded synthetic parameters;
  * every token selection of the LanguageBind / DINOv2 / ViViT registry keys is wired (a slice, a concatenation or a mean --
    merv_mean_rows -- of the encoder's full token tensor, merv_encoder_forward_select); of the SigLIP keys only
    `siglip-vit-b16-224px-all-no-cls`: the others return timm's attention-pooled (MAP head) feature per frame
    (siglip.py:46-63 leaves the timm forward in place unless the id says all-no-cls) or, for `classemb-at-first`, fail inside
    the reference itself (a tuple is reshaped, siglip.py:148-149) -- they raise NotImplementedError when constructed;
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
    """siglip.py:35-174 -- timm vit_base_patch16_siglip_224, get_intermediate_layers(n={L-2}), no class token."""

    def __init__(self, video_backbone_id: str, image_resize_strategy: str, default_image_size: int = 224,
                 num_frames: int = 8, weights=None, device="cuda:0", layers: int = 11, ln_fold: bool = True) -> None:
        super().__init__(video_backbone_id, image_resize_strategy, default_image_size, num_frames)
        if video_backbone_id != "siglip-vit-b16-224px-all-no-cls":
            raise NotImplementedError(
                f"`{video_backbone_id}`: only `siglip-vit-b16-224px-all-no-cls` is wired. Ids without `all-no-cls` keep timm's forward() "
                "(siglip.py:46-63): the attention-pooled MAP-head feature of every frame, which this path does not implement; "
                "`classemb-at-first` reshapes a tuple in the reference itself (siglip.py:148-149)")
        spec = EncoderSpec("siglip", 768, 12, 3072, layers, 16, 1, default_image_size, num_frames, "BFCHW", 0, False, False,
                           False, False, 0, "gelu_erf", 1e-6)
        self._build(spec, weights, device, lambda sd: W.from_timm_vit(sd, n_layers=layers), ln_fold)

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
