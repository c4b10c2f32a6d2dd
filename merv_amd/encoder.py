"""
HipEncoder: one visual encoder resident on one MI355X -- descriptor + device weights + stream-ordered forward
through libmerv_hip.so (merv_encoder_forward).  The four MERV encoders are instances of one parameterised
pre-LN ViT; `EncoderSpec` carries the knobs.

Canonical weight dict (fp32 CPU tensors; produced by merv_amd.weights from timm / HF / LanguageBind
state dicts, or synthetic):
    patch_w [D, 3*tt*p*p]   patch_b [D]?      prefix [npre, D]?   pos [P, D]
    pre_ln_w/b?  final_ln_w/b?
    layers[i]: ln1_w ln1_b qkv_w[3D,D] qkv_b proj_w proj_b ls1? ln2_w ln2_b fc1_w fc1_b fc2_w fc2_b ls2?
               t_emb[t,D]? t_ln_w? t_ln_b? t_qkv_w? t_qkv_b? t_proj_w? t_proj_b?
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import ACT, DT_BF16, DT_F32, PIX_LAYOUT, EncoderDesc, EncoderWeights, LayerWeights, check, ptr


@dataclass
class EncoderSpec:
    name: str
    dim: int
    heads: int
    mlp_dim: int
    layers: int            # blocks actually run (the reference computes one more and discards it)
    patch: int
    tubelet: int
    img: int
    frames: int
    pix_layout: str        # "BFCHW" | "BCFHW"
    prefix_tokens: int
    joint_space_time: bool
    pre_ln: bool
    final_ln: bool
    layerscale: bool
    temporal_frames: int
    act: str
    ln_eps: float

    @property
    def hp(self) -> int:
        return self.img // self.patch

    @property
    def t_out(self) -> int:          # VideoBackbone.temporal_resolution (base_video.py:98-101)
        return self.frames // self.tubelet

    @property
    def s_out(self) -> int:          # VideoBackbone.spatial_resolution
        return self.hp * self.hp

    @property
    def num_patches(self) -> int:
        return self.t_out * self.s_out

    @property
    def k_true(self) -> int:
        return 3 * self.tubelet * self.patch * self.patch

    @property
    def k_pad(self) -> int:
        return (self.k_true + 63) // 64 * 64

    def pixel_shape(self, batch: int):
        if self.pix_layout == "BCFHW":
            return (batch, 3, self.frames, self.img, self.img)
        return (batch, self.frames, 3, self.img, self.img)

    def flops_per_video(self) -> float:
        """Algorithmic FLOPs (2*MACs) of the consumed layers, per video (SURVEY.md section 8a accounting)."""
        D, Mh = self.dim, self.mlp_dim
        P = self.s_out * (self.t_out if self.joint_space_time else 1)
        ntok = self.prefix_tokens + P
        nseq = 1 if self.joint_space_time else self.t_out
        M = nseq * ntok
        fl = 2.0 * nseq * P * self.k_true * D  # patch embed
        per_layer = 2.0 * M * D * (3 * D + D) + 2.0 * M * D * Mh * 2 + 4.0 * nseq * ntok * ntok * D
        if self.temporal_frames:
            t = self.temporal_frames
            per_layer += 2.0 * M * D * (3 * D + D) + 4.0 * (M // t) * t * t * D
        return fl + per_layer * self.layers


def merv_full_specs() -> List[EncoderSpec]:
    """merv-full / merv-frozen encoders in registry order (merv/conf/models.py:106-113,118)."""
    return [
        EncoderSpec("languagebind", 1024, 16, 4096, 23, 14, 1, 224, 16, "BCFHW", 1, False, True, False, False, 8,
                    "gelu_erf", 1e-5),
        EncoderSpec("dinov2", 1024, 16, 4096, 23, 14, 1, 224, 16, "BFCHW", 5, False, False, False, True, 0,
                    "gelu_erf", 1e-6),
        EncoderSpec("vivit", 768, 12, 3072, 12, 16, 2, 224, 32, "BFCHW", 1, True, False, True, False, 0,
                    "gelu_tanh", 1e-6),
        EncoderSpec("siglip", 768, 12, 3072, 11, 16, 1, 224, 16, "BFCHW", 0, False, False, False, False, 0,
                    "gelu_erf", 1e-6),
    ]


_LAYER_BF16 = ("qkv_w", "proj_w", "fc1_w", "fc2_w", "t_qkv_w", "t_proj_w")


class HipEncoder:
    """Owns the device copies of one encoder's weights and the library handle."""

    def __init__(self, spec: EncoderSpec, weights: Dict, device: torch.device, ln_fold: bool = True):
        """`ln_fold` (default): LN1 / LN2 of every block are folded into the qkv / fc1 GEMMs (exact algebra) and their row
        statistics come from the epilogue of the GEMM that wrote the residual stream, so no LayerNorm pass re-reads it.
        Measured at full depth against the oracle the folded path's error (8.2e-3 fused) stays below the PyTorch-ROCm bf16
        stack's own (9.0e-3): profiles/r02_parity_calibration.json. ln_fold=False keeps separate LayerNorm kernels."""
        self.spec = spec
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipEncoder needs a ROCm device (cuda:N); merv_amd has no CPU path")
        self._lib = _lib.load()
        self._keep: List[torch.Tensor] = []
        D = spec.dim

        def dev(t: Optional[torch.Tensor], dtype) -> Optional[torch.Tensor]:
            if t is None:
                return None
            out = t.detach().to(device=self.device, dtype=dtype).contiguous()
            self._keep.append(out)
            return out

        pw = weights["patch_w"].reshape(D, -1)
        if pw.shape[1] != spec.k_true:
            raise ValueError(f"{spec.name}: patch_w has K={pw.shape[1]}, expected {spec.k_true}")
        pw_pad = torch.zeros(D, spec.k_pad, dtype=pw.dtype, device=pw.device)
        pw_pad[:, : spec.k_true] = pw
        P = spec.s_out * (spec.t_out if spec.joint_space_time else 1)
        if tuple(weights["pos"].shape) != (P, D):
            raise ValueError(f"{spec.name}: pos has shape {tuple(weights['pos'].shape)}, expected {(P, D)}")

        ew = EncoderWeights()
        ew.patch_w = ptr(dev(pw_pad, torch.bfloat16))
        ew.patch_b = ptr(dev(weights.get("patch_b"), torch.float32))
        ew.prefix = ptr(dev(weights.get("prefix"), torch.bfloat16))
        ew.pos = ptr(dev(weights["pos"], torch.bfloat16))
        ew.pre_ln_w = ptr(dev(weights.get("pre_ln_w"), torch.float32))
        ew.pre_ln_b = ptr(dev(weights.get("pre_ln_b"), torch.float32))
        ew.final_ln_w = ptr(dev(weights.get("final_ln_w"), torch.float32))
        ew.final_ln_b = ptr(dev(weights.get("final_ln_b"), torch.float32))
        if len(weights["layers"]) < spec.layers:
            raise ValueError(f"{spec.name}: {len(weights['layers'])} layer dicts < {spec.layers} layers")
        layer_arr = (LayerWeights * max(spec.layers, 1))()
        for i in range(spec.layers):
            Lw = weights["layers"][i]
            for name, _ in LayerWeights._fields_:
                t = Lw.get(name)
                setattr(layer_arr[i], name, ptr(dev(t, torch.bfloat16 if name in _LAYER_BF16 else torch.float32)))
        ew.layers = C.cast(layer_arr, C.POINTER(LayerWeights))

        desc = EncoderDesc(
            dim=D, heads=spec.heads, mlp_dim=spec.mlp_dim, layers=spec.layers, patch=spec.patch, tubelet=spec.tubelet,
            img=spec.img, frames=spec.frames, pix_layout=PIX_LAYOUT[spec.pix_layout], prefix_tokens=spec.prefix_tokens,
            joint_space_time=int(spec.joint_space_time), pre_ln=int(spec.pre_ln), final_ln=int(spec.final_ln),
            layerscale=int(spec.layerscale), temporal_frames=spec.temporal_frames, act=ACT[spec.act], k_pad=spec.k_pad,
            ln_eps=spec.ln_eps,
        )
        handle = C.c_void_p()
        check(self._lib.merv_encoder_create(C.byref(desc), C.byref(ew), C.byref(handle)), "merv_encoder_create")
        self._handle = handle
        self._ws: Optional[torch.Tensor] = None
        self._retired: List[torch.Tensor] = []
        self.ln_fold = False
        if ln_fold and spec.layers > 0:
            self.enable_ln_fold()

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            self._lib.merv_encoder_destroy(h)
            self._handle = None

    def enable_ln_fold(self) -> "HipEncoder":
        """Fold LN1 into qkv and LN2 into fc1 (exact algebra, include/merv_hip.h): the GEMMs read the residual stream directly
        and the row statistics are by-products of the producing GEMM's epilogue."""
        if self.ln_fold:
            return self
        need = self._lib.merv_encoder_ln_fold_bytes(self._handle)
        self._fold_buf = torch.empty(need, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.merv_encoder_enable_ln_fold(self._handle, ptr(self._fold_buf), need,
                                                        torch.cuda.current_stream(self.device).cuda_stream), "merv_encoder_enable_ln_fold")
        self._ws = None
        self.ln_fold = True
        return self

    def set_latency_critical(self, critical: bool) -> "HipEncoder":
        """Orchestration hint (merv_encoder_set_latency_critical): True (the default) = this encoder's chain ends its step and takes the fast wide
        form for its sub-round GEMM launches; False = it runs beside a longer chain and keeps the form that leaves that chain the CUs. Same bits."""
        if hasattr(self._lib, "merv_encoder_set_latency_critical"):  # (an A/B library of an earlier round lacks the entry: its one form stays)
            check(self._lib.merv_encoder_set_latency_critical(self._handle, 1 if critical else 0), "merv_encoder_set_latency_critical")
        self.latency_critical = bool(critical)
        return self

    MX_GEMMS = {"qkv": 1, "proj": 2, "fc1": 4, "fc2": 8}

    def enable_mxfp8(self, gemms=("qkv", "proj", "fc1", "fc2")) -> "HipEncoder":
        """Switch block GEMMs to MXFP8 operands (BASELINE.json configs[4]; include/merv_hip.h). `gemms` selects which of
        qkv / proj / fc1 / fc2 (the rest stay bf16). The library quantises the weights into a buffer this object keeps
        alive. Not the default: trades the bf16 tolerance for speed."""
        need = self._lib.merv_encoder_mxfp8_bytes(self._handle)
        self._mx_buf = torch.empty(need, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.merv_encoder_enable_mxfp8(self._handle, ptr(self._mx_buf), need,
                                                      torch.cuda.current_stream(self.device).cuda_stream), "merv_encoder_enable_mxfp8")
        mask = sum(self.MX_GEMMS[g] for g in set(gemms))
        check(self._lib.merv_encoder_set_mxfp8_mask(self._handle, mask), "merv_encoder_set_mxfp8_mask")
        self._ws = None  # the workspace grows
        self.mxfp8 = True
        return self

    def workspace(self, batch: int) -> torch.Tensor:
        need = self._lib.merv_encoder_workspace_bytes(self._handle, batch)
        if self._ws is None or self._ws.numel() < need:
            if self._ws is not None and torch.cuda.is_current_stream_capturing():
                raise RuntimeError(f"{self.spec.name}: workspace would grow inside a graph capture; warm up this batch size first")
            if self._ws is not None:
                self._retired.append(self._ws)  # a side stream may still be running in it: never hand it back to the allocator
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def pixel_shape(self, batch: int, frames: Optional[int] = None):
        s = self.spec
        f = s.frames if frames is None else frames
        return (batch, 3, f, s.img, s.img) if s.pix_layout == "BCFHW" else (batch, f, 3, s.img, s.img)

    def forward(self, pixels: torch.Tensor, out: Optional[torch.Tensor] = None,
                stream: Optional[torch.cuda.Stream] = None, frames: Optional[int] = None, select: str = "patches") -> torch.Tensor:
        """pixels in the spec's layout (fp32 or bf16, contiguous, on this device) -> [B, num_patches, D] bf16.
        `frames`: run on that many frames per video instead of the spec's count (merv_encoder_forward_frames: a
        frame-range unit; per-frame encoders only, LanguageBind in whole clips).
        `select="all"`: every token of every sequence, prefix tokens first (merv_encoder_forward_select, MERV_OUT_ALL):
        [B, sequences * (prefix + patches per sequence), D] -- the tensor the registry's other token selections slice."""
        spec = self.spec
        if pixels.device != self.device:
            raise ValueError(f"{spec.name}: pixels on {pixels.device}, encoder on {self.device}")
        B = pixels.shape[0]
        if tuple(pixels.shape) != self.pixel_shape(B, frames):
            raise ValueError(f"{spec.name}: pixel shape {tuple(pixels.shape)} != {self.pixel_shape(B, frames)}")
        if pixels.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"{spec.name}: pixels must be fp32 or bf16")
        pixels = pixels.contiguous()
        f = spec.frames if frames is None else frames
        n_out = (f // spec.tubelet) * spec.s_out
        if select == "all":
            n_out += spec.prefix_tokens * (1 if spec.joint_space_time else f // spec.tubelet)
        elif select != "patches":
            raise ValueError(f"{spec.name}: select must be 'patches' or 'all', got {select!r}")
        with torch.cuda.device(self.device):  # the library launches on the CURRENT HIP device: make it this encoder's
            if out is None:
                out = torch.empty(B, n_out, spec.dim, dtype=torch.bfloat16, device=self.device)
            elif tuple(out.shape) != (B, n_out, spec.dim) or out.dtype != torch.bfloat16 or not out.is_contiguous():
                raise ValueError(f"{spec.name}: out must be a contiguous bf16 {(B, n_out, spec.dim)} tensor")
            ws = self.workspace(B)
            s = stream if stream is not None else torch.cuda.current_stream(self.device)
            rc = self._lib.merv_encoder_forward_select(
                self._handle, ptr(pixels), DT_BF16 if pixels.dtype == torch.bfloat16 else DT_F32, B, f, 1 if select == "all" else 0,
                ptr(out), ptr(ws), ws.numel(), s.cuda_stream)
        check(rc, f"merv_encoder_forward[{spec.name}]")
        return out
