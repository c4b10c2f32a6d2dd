"""
Projector and fusion modules of the MERV visual path, with the reference's class names, constructor arguments and
state-dict keys (merv/util/nn_utils.py), so reference checkpoints (`projectors.*`, `feature_fusion.*`,
merv/models/vidlms/merv.py:272-289) load unchanged -- but whose forward() runs the HIP kernels of libmerv_hip.so.

  LinearProjector                      nn_utils.py:22-32
  AveragePooling3DProjector            nn_utils.py:306-338   (mlp_type "linear" only: the `3davg+linear` arch)
  CrossAttentionAdapterLearnableQuery  nn_utils.py:455-521   (averagetoken=True only: `cross_attention_avg_lq`)

Inference only (the encoders are frozen in every reference stage and these modules are called under no_grad in
generate()); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
from torch.nn.init import xavier_uniform_
from torch.nn.parameter import Parameter

from . import _lib
from ._lib import check, ptr


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class LinearProjector(nn.Module):
    def __init__(self, vision_dim: int, llm_dim: int, pre_proj_layernorm: bool = False) -> None:
        super().__init__()
        if pre_proj_layernorm:
            raise NotImplementedError("pre_proj_layernorm is not used by the 3davg+linear arch (merv.py:152-163)")
        self.projector = nn.Linear(vision_dim, llm_dim, bias=True)
        self.layernorm = nn.Identity()


class AveragePooling3DProjector(nn.Module):
    """3D-average pooling projector: [B, F, N, C] -> AdaptiveAvgPool3d((output_frames, s, s)) -> Linear -> [B, F*s*s, llm]."""

    def __init__(self, fused_vision_dim: int, llm_dim: int, output_frames: int, output_size: int,
                 mlp_type: str = "gelu-mlp") -> None:
        super().__init__()
        if mlp_type != "linear":
            raise ValueError(f"Projector with `{mlp_type = }` is not supported by the HIP path (only 'linear')")
        self.output_frames = output_frames
        self.output_size = output_size
        self.projector = LinearProjector(fused_vision_dim, llm_dim)
        self._dev = None  # (weight bf16, bias fp32) device copies

    @property
    def output_token_length(self) -> int:
        return self.output_size * self.output_size

    @property
    def output_frame_length(self) -> int:
        return self.output_frames

    def prepare(self, device) -> None:
        lin = self.projector.projector
        self._dev = (lin.weight.detach().to(device=device, dtype=torch.bfloat16).contiguous(),
                     lin.bias.detach().to(device=device, dtype=torch.float32).contiguous())

    def forward(self, fused_img_patches: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        # nn_utils.py:320-330
        assert fused_img_patches.dim() == 4, "expected [B, F, N, C]"
        B, Fr, N, Cc = fused_img_patches.shape
        H = int(math.sqrt(N))
        if H * H != N:
            raise ValueError(f"spatial token count {N} is not a square")
        if Fr != self.output_frames:
            raise NotImplementedError("temporal pooling (output_frames != input frames) is not used by merv-full")
        x = fused_img_patches
        if not x.is_cuda:
            raise RuntimeError("AveragePooling3DProjector: HIP path needs a ROCm device tensor (no CPU fallback)")
        if x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
        x = x.contiguous()
        if self._dev is None or self._dev[0].device != x.device:
            self.prepare(x.device)
        w, b = self._dev
        llm = w.shape[0]
        T, o = Fr, self.output_size
        with torch.cuda.device(x.device):
            pooled = torch.empty(B * T * o * o, Cc, dtype=torch.bfloat16, device=x.device)
            if out is None:
                out = torch.empty(B, T * o * o, llm, dtype=torch.bfloat16, device=x.device)
            rc = _lib.load().merv_projector_forward(ptr(x), B, T, H, Cc, o, ptr(w), ptr(b), llm, ptr(pooled), ptr(out),
                                                    _stream_ptr(x.device))
        check(rc, "merv_projector_forward")
        return out


class CrossAttentionAdapterLearnableQuery(nn.Module):
    def __init__(self, embed_dim=3072, llm_dim=4098, token_length=8, averagetoken=False, num_encoder=4,
                 positional_embedding=False) -> None:
        super().__init__()
        if not averagetoken or positional_embedding:
            raise NotImplementedError("only averagetoken=True, positional_embedding=False (cross_attention_avg_lq)")
        self.llm_dim = llm_dim
        self.token_length = token_length
        self.averagetoken = averagetoken
        # same parameter names / shapes as nn.MultiheadAttention(embed_dim, 1, kdim=vdim=llm_dim) so that the
        # reference's `feature_fusion` state dict loads; v_proj / out_proj only feed the discarded output (:512)
        self.attention = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=1, dropout=0.0, batch_first=True,
                                               kdim=llm_dim, vdim=llm_dim)
        self.Q = Parameter(torch.empty((1, embed_dim)))
        self.num_encoder = num_encoder
        self.positional_embedding = positional_embedding
        xavier_uniform_(self.Q)
        self._u = None

    def fold(self) -> torch.Tensor:
        """u = Wk^T (Wq Q + bq) / sqrt(embed_dim), fp64 on the host, so that score_e = mean_t(V_e) . u (+ const)."""
        a = self.attention
        Ed = self.Q.shape[1]
        bq = a.in_proj_bias.detach().double()[:Ed]
        q = a.q_proj_weight.detach().double() @ self.Q.detach().double()[0] + bq
        return (a.k_proj_weight.detach().double().t() @ q / math.sqrt(Ed)).float()

    def prepare(self, device) -> None:
        self._u = self.fold().to(device).contiguous()

    def forward(self, V: Sequence[torch.Tensor], out: Optional[torch.Tensor] = None, partial: Optional[torch.Tensor] = None,
                weights: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """nn_utils.py:487-521 -> (sum_e w_e V_e [B,T,C] bf16, w [B,E] fp32). `out` / `partial` / `weights`: caller-owned
        result and scratch buffers (MervVisualPath keeps them persistent); allocated here when omitted."""
        for emb in V:  # nn_utils.py:494-495
            assert emb.shape[1] == self.token_length or emb.shape[1] == 1, (self.token_length, [e.shape for e in V])
        V = [(emb.repeat(1, self.token_length, 1) if emb.shape[1] == 1 else emb) for emb in V]
        E = len(V)
        B, T, Cc = V[0].shape
        dev = V[0].device
        if dev.type != "cuda":
            raise RuntimeError("CrossAttentionAdapterLearnableQuery: HIP path needs ROCm device tensors")
        Vc = [v.to(torch.bfloat16).contiguous() for v in V]
        if self._u is None or self._u.device != dev:
            self.prepare(dev)
        lib = _lib.load()
        need = lib.merv_fusion_workspace_floats(B, E, T)
        with torch.cuda.device(dev):
            if partial is None:
                partial = torch.empty(need, dtype=torch.float32, device=dev)
            elif partial.numel() < need or partial.dtype != torch.float32:
                raise ValueError("fusion: partial workspace too small")
            if weights is None:
                weights = torch.empty(B, E, dtype=torch.float32, device=dev)
            if out is None:
                out = torch.empty(B, T, Cc, dtype=torch.bfloat16, device=dev)
            arr = (C.c_void_p * E)(*[ptr(v) for v in Vc])
            rc = lib.merv_fusion_forward(arr, E, B, T, Cc, ptr(self._u), ptr(partial), ptr(weights), ptr(out),
                                         _stream_ptr(dev))
        check(rc, "merv_fusion_forward")
        return out, weights


def splice(input_embeddings: torch.Tensor, fused: torch.Tensor, bos_token_length: int = 1) -> torch.Tensor:
    """merv.py:633-640: cat[emb[:, :bos], fused, emb[:, bos:]] as one HIP copy kernel (bf16)."""
    emb = input_embeddings.to(torch.bfloat16).contiguous()
    vis = fused.to(torch.bfloat16).contiguous()
    B, S, Cc = emb.shape
    T = vis.shape[1]
    with torch.cuda.device(emb.device):
        out = torch.empty(B, S + T, Cc, dtype=torch.bfloat16, device=emb.device)
        rc = _lib.load().merv_splice_forward(ptr(emb), ptr(vis), B, S, T, Cc, bos_token_length, ptr(out),
                                             _stream_ptr(emb.device))
    check(rc, "merv_splice_forward")
    return out
