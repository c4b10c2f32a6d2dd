"""
Per-encoder video transforms on the GPU, bound to libmerv_hip.so:

  PILResizeNormalize   <- Compose([to_pil_image, Resize((224,224), interp), CenterCrop(224), ToTensor(), Normalize(mean,std)])
                          applied per frame and stacked (dinov2_video.py:123-124, siglip.py:133-134, vivit.py:91-92);
                          Resize((224,224)) makes the CenterCrop a no-op ("resize-naive", conf/models.py:115)
  LanguageBindTransform <- get_video_transform() of languagebind/video/processing_video.py:63-79, applied to
                          video.permute(1,0,2,3) (languagebind/__init__.py:71)

Input: uint8 frames [T, 3, H, W] (load_video's layout) on the ROCm device. The resized uint8 image of the PIL path is
bit-exact with Pillow; the float stages follow the reference's operation order with one fp32 rounding per operation.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import DT_BF16, DT_F32, check, ptr

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)      # timm DINOv2 data config
HALF_MEAN, HALF_STD = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)                          # timm SigLIP / VivitImageProcessor
OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)                        # processing_video.py:23-24
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)
FILTERS = {"bilinear": 0, "bicubic": 1}


def _f3(v: Sequence[float]):
    return (C.c_float * 3)(*[float(x) for x in v])


def _check_frames(video: torch.Tensor):
    if not (video.is_cuda and video.dtype == torch.uint8 and video.dim() == 4 and video.shape[1] == 3):
        raise ValueError("expected uint8 frames [T, 3, H, W] on a ROCm device (load_video's output layout)")
    return video.contiguous()


class PILResizeNormalize:
    """video uint8 [T,3,H,W] -> [T,3,size,size]; `dtype` fp32 (the reference's) or bf16 (what the encoders consume)."""

    def __init__(self, interpolation: str, mean, std, size: int = 224, dtype: torch.dtype = torch.float32):
        self.filter = FILTERS[interpolation]
        self.mean, self.std, self.size, self.dtype = tuple(mean), tuple(std), size, dtype

    def __call__(self, video: torch.Tensor, return_uint8: bool = False):
        video = _check_frames(video)
        T, _, H, W = video.shape
        lib = _lib.load()
        S = self.size
        ws = torch.empty(lib.merv_preprocess_workspace_bytes(T, H, W, S), dtype=torch.uint8, device=video.device)
        out = torch.empty(T, 3, S, S, dtype=self.dtype, device=video.device)
        u8 = torch.empty(T, 3, S, S, dtype=torch.uint8, device=video.device) if return_uint8 else None
        rc = lib.merv_preprocess_pil(ptr(video), T, H, W, S, self.filter, _f3(self.mean), _f3(self.std), ptr(out),
                                     DT_BF16 if self.dtype == torch.bfloat16 else DT_F32, ptr(u8), ptr(ws), ws.numel(),
                                     torch.cuda.current_stream(video.device).cuda_stream)
        check(rc, "merv_preprocess_pil")
        return (out, u8) if return_uint8 else out


class LanguageBindTransform:
    """video uint8 [T,3,H,W] -> [3,T,224,224] (the [C,T,H,W] layout LangBindVideoBackbone.forward expects per sample)."""

    def __init__(self, size: int = 224, dtype: torch.dtype = torch.float32, flip: bool = False):
        self.size, self.dtype, self.flip = size, dtype, flip

    def __call__(self, video: torch.Tensor) -> torch.Tensor:
        video = _check_frames(video)
        T, _, H, W = video.shape
        out = torch.empty(3, T, self.size, self.size, dtype=self.dtype, device=video.device)
        rc = _lib.load().merv_preprocess_languagebind(ptr(video), T, H, W, self.size, int(self.flip), _f3(OPENAI_DATASET_MEAN),
                                                      _f3(OPENAI_DATASET_STD), ptr(out),
                                                      DT_BF16 if self.dtype == torch.bfloat16 else DT_F32,
                                                      torch.cuda.current_stream(video.device).cuda_stream)
        check(rc, "merv_preprocess_languagebind")
        return out


def transform_for(encoder_name: str, dtype: torch.dtype = torch.float32):
    """The reference's video_transform of each merv-full backbone."""
    if encoder_name == "languagebind":
        return LanguageBindTransform(dtype=dtype)
    if encoder_name == "dinov2":
        return PILResizeNormalize("bicubic", IMAGENET_MEAN, IMAGENET_STD, dtype=dtype)
    if encoder_name == "siglip":
        return PILResizeNormalize("bicubic", HALF_MEAN, HALF_STD, dtype=dtype)
    if encoder_name == "vivit":
        return PILResizeNormalize("bilinear", HALF_MEAN, HALF_STD, dtype=dtype)
    raise ValueError(encoder_name)
