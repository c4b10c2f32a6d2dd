"""
Tensor-level wrappers of the single-kernel C-ABI entry points (include/merv_hip.h, "single kernels" section).
Used by the parity tests and micro-benchmarks; the encoders themselves run through merv_encoder_forward.
All tensors must be contiguous CUDA (ROCm) tensors: bf16 activations / GEMM weights, fp32 vectors.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import ACT, DT_BF16, DT_F32, check, current_stream_ptr, ptr


def _chk(t: Optional[torch.Tensor], dtype, name: str):
    if t is None:
        return
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous CUDA tensor of {dtype}, got {t.dtype} on {t.device}")


def gemm(a: torch.Tensor, w: torch.Tensor, bias=None, act: str = "none", lscale=None, res=None, res_row_mod: int = 0,
         out: Optional[torch.Tensor] = None, out_group: int = 0, out_stride: int = 0, out_off: int = 0) -> torch.Tensor:
    """out = res + lscale * act(a @ w.T + bias); a [M,K] bf16, w [N,K] bf16 (nn.Linear layout)."""
    _chk(a, torch.bfloat16, "a"); _chk(w, torch.bfloat16, "w"); _chk(bias, torch.float32, "bias")
    _chk(lscale, torch.float32, "lscale"); _chk(res, torch.bfloat16, "res")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"gemm: K mismatch {a.shape} x {w.shape}")
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a.device)
    _chk(out, torch.bfloat16, "out")
    lib = _lib.load()
    rc = lib.merv_gemm_bf16(ptr(a), ptr(w), ptr(out), ptr(bias), ptr(lscale), ptr(res), M, N, K, K, K, out.shape[-1],
                            res.shape[-1] if res is not None else 0, res_row_mod, out_group, out_stride, out_off,
                            ACT[act], current_stream_ptr(a.device))
    check(rc, "merv_gemm_bf16")
    return out


def layernorm(x: torch.Tensor, gamma, beta, eps: float, add=None, add_div: int = 1, add_mod: int = 1) -> torch.Tensor:
    """y = LN(x (+ add[(row // add_div) % add_mod], written back into x)). x [M,D] bf16."""
    _chk(x, torch.bfloat16, "x"); _chk(gamma, torch.float32, "gamma"); _chk(beta, torch.float32, "beta")
    _chk(add, torch.float32, "add")
    M, D = x.shape
    y = torch.empty_like(x)
    rc = _lib.load().merv_layernorm(ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(add), M, D, add_div, add_mod, eps,
                                    current_stream_ptr(x.device))
    check(rc, "merv_layernorm")
    return y


def attention(qkv: torch.Tensor, nseq: int, L: int, heads: int) -> torch.Tensor:
    """qkv [nseq*L, 3*D] bf16 -> [nseq*L, D] bf16, head_dim 64."""
    _chk(qkv, torch.bfloat16, "qkv")
    D = heads * 64
    if qkv.shape != (nseq * L, 3 * D):
        raise ValueError(f"attention: qkv shape {tuple(qkv.shape)} != {(nseq * L, 3 * D)}")
    out = torch.empty(nseq * L, D, dtype=torch.bfloat16, device=qkv.device)
    rc = _lib.load().merv_attention(ptr(qkv), ptr(out), nseq, L, heads, D, 0.125, current_stream_ptr(qkv.device))
    check(rc, "merv_attention")
    return out


def temporal_attention(qkv: torch.Tensor, nclips: int, t: int, ntok: int, heads: int) -> torch.Tensor:
    """Rows are frame-major (row = (clip*t + i) * ntok + token); attends over i for each (clip, token, head)."""
    _chk(qkv, torch.bfloat16, "qkv")
    D = heads * 64
    if qkv.shape != (nclips * t * ntok, 3 * D):
        raise ValueError("temporal_attention: bad qkv shape")
    out = torch.empty(nclips * t * ntok, D, dtype=torch.bfloat16, device=qkv.device)
    rc = _lib.load().merv_temporal_attention(ptr(qkv), ptr(out), nclips, t, ntok, heads, D, 0.125,
                                             current_stream_ptr(qkv.device))
    check(rc, "merv_temporal_attention")
    return out


def im2col(pix: torch.Tensor, layout: str, patch: int, tubelet: int, k_pad: int) -> torch.Tensor:
    """pix [B,F,3,H,W] ("BFCHW") or [B,3,F,H,W] ("BCFHW"), fp32 or bf16 -> [B*(F/tubelet)*hp*hp, k_pad] bf16."""
    if not pix.is_cuda or not pix.is_contiguous() or pix.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("im2col: contiguous CUDA fp32/bf16 pixels required")
    if layout == "BFCHW":
        B, Fr, _, H, W = pix.shape
        sB, sF, sC = Fr * 3 * H * W, 3 * H * W, H * W
    else:
        B, _, Fr, H, W = pix.shape
        sB, sC, sF = 3 * Fr * H * W, Fr * H * W, H * W
    hp = H // patch
    out = torch.empty(B * (Fr // tubelet) * hp * hp, k_pad, dtype=torch.bfloat16, device=pix.device)
    rc = _lib.load().merv_im2col(ptr(pix), DT_BF16 if pix.dtype == torch.bfloat16 else DT_F32, ptr(out), B, Fr, H, patch,
                                 tubelet, k_pad, sB, sF, sC, current_stream_ptr(pix.device))
    check(rc, "merv_im2col")
    return out


def pool3d(tokens: torch.Tensor, T: int, S: int, out_size: int) -> torch.Tensor:
    """tokens [B, T*S*S, C] bf16 -> [B, T*out*out, C] bf16 (AdaptiveAvgPool3d((T,out,out)))."""
    _chk(tokens, torch.bfloat16, "tokens")
    B, N, Cc = tokens.shape
    if N != T * S * S:
        raise ValueError("pool3d: token count != T*S*S")
    out = torch.empty(B, T * out_size * out_size, Cc, dtype=torch.bfloat16, device=tokens.device)
    rc = _lib.load().merv_pool3d(ptr(tokens), ptr(out), B, T, S, out_size, Cc, current_stream_ptr(tokens.device))
    check(rc, "merv_pool3d")
    return out


# ---- MXFP8 mode (BASELINE.json configs[4]) ----
def quantize_mxfp8(x: torch.Tensor):
    """bf16 [rows, K] -> (q uint8 [rows, K] OCP e4m3, scales uint8 in the GEMM's lane order); include/merv_hip.h."""
    lib = _lib.load()
    if x.dtype != torch.bfloat16 or x.dim() != 2 or x.stride(1) != 1 or not x.is_cuda:
        raise ValueError("quantize_mxfp8: expected a CUDA bf16 [rows, K] tensor with contiguous rows")
    rows, K = x.shape
    if K % 128:
        raise ValueError("quantize_mxfp8: K must be a multiple of 128")
    q = torch.empty(rows, K, dtype=torch.uint8, device=x.device)
    sc = torch.zeros(lib.merv_mxfp8_scale_bytes(rows, K), dtype=torch.uint8, device=x.device)
    check(lib.merv_quantize_mxfp8(ptr(x), rows, K, x.stride(0), ptr(q), ptr(sc), current_stream_ptr(x.device)), "merv_quantize_mxfp8")
    return q, sc


def mxfp8_scales_to_rows(sc: torch.Tensor, rows: int, K: int) -> torch.Tensor:
    """Un-permute the lane-order scale array to [rows, K/32] (tests / inspection)."""
    groups = (rows + 63) // 64
    s = sc.view(K // 128, groups, 4, 16, 4)  # [ktile][group][kblock%4][row%16][(row%64)/16]
    s = s.permute(1, 4, 3, 0, 2).reshape(groups * 64, K // 32)  # row = 64*group + 16*j + r ; kblock = 4*ktile + kq
    return s[:rows]


def gemm_mxfp8(aq, a_sc, wq, w_sc, bias=None, act="none", lscale=None, res=None, out=None):
    lib = _lib.load()
    M, K = aq.shape
    N = wq.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=aq.device)
    check(lib.merv_gemm_mxfp8(ptr(aq), ptr(a_sc), ptr(wq), ptr(w_sc), ptr(out), ptr(bias), ptr(lscale), ptr(res), M, N, K,
                              aq.stride(0), wq.stride(0), out.stride(0), res.stride(0) if res is not None else 0, ACT[act],
                              current_stream_ptr(aq.device)), "merv_gemm_mxfp8")
    return out
