"""
ctypes binding of libmerv_hip.so (C ABI: include/merv_hip.h).

The product path has no CPU fallback: if the shared library is missing or does not export the ABI, importing a
symbol from here raises immediately.  `import torch` happens first on purpose -- torch ships its own
libamdhip64.so (SONAME libamdhip64.so.7); loading it first makes this library bind to the same HIP runtime, so
device pointers and streams are interchangeable.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from pathlib import Path

import torch  # noqa: F401  (must precede CDLL: shares the HIP runtime)

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libmerv_hip.so"
# the test / probe build of the same sources (-DMERV_TUNING_HOOKS, `make hooks`): loaded instead of the product library only when
# MERV_TUNING_HOOKS=1 is set. The product library reads no environment variable (no hidden global state).
HOOKS_LIB_PATH = LIB_PATH.with_name("libmerv_hip_hooks.so")


def tuning_hooks_enabled() -> bool:
    return os.environ.get("MERV_TUNING_HOOKS") == "1"


def tuning(name: str, default=None):
    """A MERV_* tuning / A-B variable -- honoured only under MERV_TUNING_HOOKS=1 (tests, tools/probes); otherwise `default`: a host
    process's environment does not change what the product path runs."""
    return os.environ.get(name, default) if tuning_hooks_enabled() else default


ACT = {"none": 0, "gelu_erf": 1, "gelu_tanh": 2, "quick_gelu": 3}
PIX_LAYOUT = {"BFCHW": 0, "BCFHW": 1}
DT_F32, DT_BF16 = 0, 1


class EncoderDesc(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("heads", C.c_int32), ("mlp_dim", C.c_int32), ("layers", C.c_int32),
        ("patch", C.c_int32), ("tubelet", C.c_int32), ("img", C.c_int32), ("frames", C.c_int32),
        ("pix_layout", C.c_int32), ("prefix_tokens", C.c_int32), ("joint_space_time", C.c_int32),
        ("pre_ln", C.c_int32), ("final_ln", C.c_int32), ("layerscale", C.c_int32),
        ("temporal_frames", C.c_int32), ("act", C.c_int32), ("k_pad", C.c_int32), ("ln_eps", C.c_float),
    ]


_LAYER_FIELDS = [
    "ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1", "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w",
    "fc2_b", "ls2", "t_emb", "t_ln_w", "t_ln_b", "t_qkv_w", "t_qkv_b", "t_proj_w", "t_proj_b",
]


class LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _LAYER_FIELDS]


class EncoderWeights(C.Structure):
    _fields_ = [
        ("patch_w", C.c_void_p), ("patch_b", C.c_void_p), ("prefix", C.c_void_p), ("pos", C.c_void_p),
        ("pre_ln_w", C.c_void_p), ("pre_ln_b", C.c_void_p), ("final_ln_w", C.c_void_p), ("final_ln_b", C.c_void_p),
        ("layers", C.POINTER(LayerWeights)),
    ]


# name -> (restype, argtypes); every symbol include/merv_hip.h declares
_i32, _i64, _f32, _f64, _vp, _sz = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p, C.c_size_t


ABI_VERSION = 4  # include/merv_hip.h MERV_ABI_VERSION this binding was written against

SIGNATURES = {
    "merv_last_error": (C.c_char_p, []),
    "merv_abi_version": (C.c_int, []),
    "merv_frame_indices": (C.c_int, [_i64, _f64, _f64, _f64, _i64, _i32, C.POINTER(_i64)]),
    "merv_temporal_subsample": (C.c_int, [_i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "merv_encoder_create": (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(EncoderWeights), C.POINTER(_vp)]),
    "merv_encoder_destroy": (None, [_vp]),
    "merv_encoder_workspace_bytes": (_sz, [_vp, _i32]),
    "merv_encoder_num_patches": (_i32, [_vp]),
    "merv_encoder_forward": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "merv_encoder_forward_frames": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "merv_encoder_forward_select": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "merv_mean_rows": (C.c_int, [_vp, _vp, _i32, _i32, _i32, C.c_int64, _vp]),
    "merv_map_pool_attention": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "merv_projector_forward": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "merv_fusion_workspace_floats": (_sz, [_i32, _i32, _i32]),
    "merv_fusion_forward": (C.c_int, [C.POINTER(_vp), _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "merv_splice_forward": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "merv_fusion_backward_workspace_floats": (_sz, [_i32, _i32, _i32, _i32]),
    "merv_fusion_backward_reduce": (C.c_int, [C.POINTER(_vp), _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "merv_fusion_backward_mix": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, C.POINTER(_vp), _vp]),
    "merv_projector_backward_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "merv_projector_backward": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _sz, _vp, _vp, _vp]),
    "merv_encoder_ln_fold_bytes": (_sz, [_vp]),
    "merv_encoder_enable_ln_fold": (C.c_int, [_vp, _vp, _sz, _vp]),
    "merv_encoder_mxfp8_bytes": (_sz, [_vp]),
    "merv_encoder_enable_mxfp8": (C.c_int, [_vp, _vp, _sz, _vp]),
    "merv_encoder_set_mxfp8_mask": (C.c_int, [_vp, _i32]),
    "merv_encoder_set_latency_critical": (C.c_int, [_vp, _i32]),
    "merv_debug_gemm_mx_out": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "merv_debug_gemm_mxfp8_forms": (C.c_int, [_vp] * 8 + [_i32] * 4 + [_vp] * 4 + [_i32, _i32, _vp, _vp, _i32, _i32, _vp]),
    "merv_debug_gemm_stats": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "merv_debug_gemm_row_add": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp]),
    "merv_mxfp8_scale_bytes": (_sz, [_i32, _i32]),
    "merv_quantize_mxfp8": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "merv_gemm_mxfp8": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp] + [_i32] * 8 + [_vp]),
    "merv_transpose_bf16": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "merv_gemm_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [_i32] * 12 + [_vp]),
    "merv_layernorm": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "merv_attention": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "merv_temporal_attention": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "merv_im2col": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _vp]),
    "merv_pool3d": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "merv_preprocess_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "merv_preprocess_pil": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(_f32), C.POINTER(_f32), _vp, _i32, _vp, _vp, _sz, _vp]),
    "merv_preprocess_languagebind": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(_f32), C.POINTER(_f32), _vp, _i32, _vp]),
    "merv_decode_rmsnorm": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "merv_decode_gemv": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _f32, _vp]),
    "merv_decode_gemv3": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _f32, _vp]),
    "merv_decode_gemv3_bias": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _f32, _vp, _vp, _vp, _vp]),
    "merv_decode_rope_cache": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "merv_prefill_rope_cache": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "merv_silu_mul": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "merv_add_rmsnorm": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "merv_prefill_attention": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, C.c_int64, _i32, _f32, _vp]),
    "merv_decode_attention_workspace_floats": (_sz, [_i32, _i32]),
    "merv_decode_attention": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "merv_decode_attention_fused_workspace_floats": (_sz, [_i32, _i32]),
    "merv_decode_attention_fused": (C.c_int, [_vp] * 10 + [_i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "merv_decode_attention_split_workspace_floats": (_sz, [_i32, _i32]),
    "merv_decode_attention_split": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "merv_decode_attention_split_prefetch": (C.c_int, [_vp] * 9 + [_i32] * 5 + [_f32, _vp, C.c_int64, _i32, _vp]),
    "merv_decode_oproj_merge": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "merv_decode_greedy_advance": (C.c_int, [_vp, _i32, _vp, _vp, _vp, C.c_int64, _vp]),
    "merv_decode_sample_advance": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "merv_tuning_hooks": (C.c_int, []),
    "merv_debug_set_gemm_variant": (None, [_i32]),
    "merv_debug_set_attn_rescale_thr": (None, [_f32]),
    "merv_debug_set_rest_fork": (None, [_vp, _vp]),
    "merv_prof_enable": (None, [_i32]),
    "merv_prof_reset": (None, []),
    "merv_prof_read": (C.c_int, [_i32, C.POINTER(_f64), C.POINTER(_i64), C.POINTER(_f64), C.POINTER(_f64)]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library and bind every ABI symbol. Raises (never falls back) when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("MERV_HIP_LIB", HOOKS_LIB_PATH if tuning_hooks_enabled() else LIB_PATH))
    if not path.exists():
        raise RuntimeError(
            f"{path.name} not found at {path}: build it with `make lib hooks` (or __graft_entry__.build()). "
            "merv_amd has no CPU fallback."
        )
    lib = C.CDLL(str(path))
    # MERV_HIP_LIB is the ordinary library-path override and is checked as strictly as the in-tree build (every ABI symbol, the ABI
    # version). Only with MERV_HIP_LIB_AB=1 beside it (same-box A/B pairs against a previous round's build, tools/probes/ab_*.sh) do
    # symbols that library lacks stay unbound (using one raises) and is its ABI version only reported.
    ab_build = "MERV_HIP_LIB" in os.environ and os.environ.get("MERV_HIP_LIB_AB") == "1"
    for name, (res, args) in SIGNATURES.items():
        if ab_build and not hasattr(lib, name):
            print(f"[merv_amd] A/B library {path} does not export {name}", file=sys.stderr)
            continue
        fn = getattr(lib, name)  # AttributeError if the .so does not export the ABI
        fn.restype = res
        fn.argtypes = args
    if lib.merv_abi_version() != ABI_VERSION:
        if not ab_build:
            raise RuntimeError(f"libmerv_hip.so ABI version {lib.merv_abi_version()} != {ABI_VERSION} (stale build: run `make lib`)")
        print(f"[merv_amd] A/B library {path}: ABI version {lib.merv_abi_version()} (binding written against {ABI_VERSION})", file=sys.stderr)
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    """Mirror of the reference's failure modes: bad arguments -> ValueError, device/runtime errors -> RuntimeError."""
    if rc == 0:
        return
    msg = load().merv_last_error().decode("utf-8", "replace")
    if rc == 1:
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg}")


def ptr(t) -> int:
    """Raw device (or host) pointer of a tensor, or 0 for None."""
    return 0 if t is None else t.data_ptr()


def current_stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream
