"""GPU: every GEMM tile configuration (forced through merv_debug_set_gemm_variant) against torch fp32, incl. ragged M,
many tiles per persistent block, fused epilogues and in-place residual."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("hooks_library")]  # forced tile configurations: the hooks build (conftest.py)


# variant 0 = automatic choice, incl. the split between the eight-phase kernel (complete rounds) and a smaller-tile
# launch for the remaining rows; 7 = eight-phase forced (needs N % 256 == 0 and an even number >= 4 of K-tiles);
# 9 = 64 x 128 blocks (remainders of a few hundred rows)
@pytest.mark.parametrize("variant", [0, 1, 3, 4, 6, 7, 9])
@pytest.mark.parametrize("M,N,K", [(1, 256, 64), (300, 256, 128), (4112, 1024, 1024), (33000, 768, 192), (70000, 256, 64),
                                   (33000, 768, 256), (513, 512, 640), (66000, 512, 256), (25096, 2304, 768)])
def test_variant(dev, variant, M, N, K):
    from merv_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K + variant)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * K**-0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ls = (0.5 + torch.rand(N, generator=g)).to(dev)
    x = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    ref = x.float() + ls * F.gelu(a.float() @ w.float().t() + bias)
    lib.merv_debug_set_gemm_variant(variant)
    try:
        ops.gemm(a, w, bias=bias, act="gelu_erf", lscale=ls, res=x, out=x)  # in place
        plain = ops.gemm(a, w)
    finally:
        lib.merv_debug_set_gemm_variant(0)
    assert rel_l2(x, ref) < 6e-3, (variant, M, N, K)
    assert rel_l2(plain, a.float() @ w.float().t()) < 6e-3


@pytest.mark.parametrize("act,ls,res", [("none", False, False), ("none", False, True), ("none", True, True), ("gelu_erf", False, False),
                                        ("quick_gelu", False, False)])
def test_static_and_run_time_epilogue_forms_give_the_same_bits(dev, act, ls, res):
    """Whole-tile launches (M a multiple of the tile) take the static epilogue instantiations (gemm.hip, WHOLE: every optional term a template
    flag), ragged ones the run-time form. The rows of a 65536-row problem (256 whole m-tiles: static, eight-phase kernel) must equal, bit for
    bit, the same rows computed as the head of a 65537-row problem (the last tile ragged: run-time form for the rows the eight-phase kernel
    takes when the split differs, small tiles for the rest) -- whichever kernel and form a row lands in."""
    from merv_amd import ops
    g = torch.Generator().manual_seed(17)
    M, N, K = 65536, 512, 256
    a = torch.randn(M + 1, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * K**-0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    lscale = (0.5 + torch.rand(N, generator=g)).to(dev) if ls else None
    x = torch.randn(M + 1, N, generator=g).to(torch.bfloat16).to(dev) if res else None
    whole = ops.gemm(a[:M], w, bias=bias, act=act, lscale=lscale, res=None if x is None else x[:M])
    ragged = ops.gemm(a, w, bias=bias, act=act, lscale=lscale, res=x)
    assert torch.equal(whole, ragged[:M])
    # a sub-round launch (few tiles: the eight-phase kernel takes the whole ragged problem in its run-time form) against the static form
    Ms = 256 * 40
    small_whole = ops.gemm(a[:Ms], w, bias=bias, act=act, lscale=lscale, res=None if x is None else x[:Ms])
    small_ragged = ops.gemm(a[: Ms + 3], w, bias=bias, act=act, lscale=lscale, res=None if x is None else x[: Ms + 3])
    assert torch.equal(small_whole, small_ragged[:Ms]) and torch.equal(small_whole, whole[:Ms])


def test_split_launch_carries_row_offsets_of_optional_epilogue_inputs(dev):
    """A GEMM large enough for the round-filling split (eight-phase part + remaining rows): the folded-LayerNorm row
    statistics and the MXFP8 output of the SECOND launch must be offset by the rows the first one took."""
    import ctypes as C
    from merv_amd import _lib, ops
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    M, N, K = 66000, 512, 256  # 257 full m-tiles x 2 n-tiles: 2 complete rounds to the eight-phase kernel, 464 rows left
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * K**-0.5).to(torch.bfloat16).to(dev)
    plain = ops.gemm(a, w)
    q, sc = ops.quantize_mxfp8(plain)
    # MXFP8 output straight from the epilogue == quantising the bf16 result afterwards, incl. the remainder rows
    q2 = torch.empty(M, N, dtype=torch.uint8, device=dev)
    sc2 = torch.zeros_like(sc)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    check(lib.merv_debug_gemm_mx_out(ptr(a), ptr(w), ptr(out), M, N, K, ptr(q2), ptr(sc2), torch.cuda.current_stream(dev).cuda_stream),
          "merv_debug_gemm_mx_out")
    assert torch.equal(q2, q) and torch.equal(ops.mxfp8_scales_to_rows(sc2, M, N), ops.mxfp8_scales_to_rows(sc, M, N))


@pytest.mark.parametrize("M,N,K,act", [(128, 1024, 4096, "none"), (640, 1024, 1024, "none"), (33000, 1024, 256, "none"),
                                       (100, 768, 3072, "gelu_tanh"), (640, 4096, 1024, "gelu_erf"), (33, 256, 1024, "quick_gelu")])
def test_epilogue_layernorm_partials(dev, M, N, K, act):
    """The LayerNorm partials a GEMM's epilogue leaves for a LayerNorm folded into the next GEMM: {sum, M2} per 64 columns of
    the bf16-rounded output row (every tile configuration shares the epilogue; 33000 rows take the round-filling split, whose
    second launch must offset the partials by the rows the first one took). Combined by Chan's rule they must give the row
    mean / variance of what was stored; rows are given an offset so that mean >> 0."""
    import torch.nn.functional as F
    from merv_amd import _lib
    from merv_amd._lib import ACT, check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * K**-0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ls = (0.5 + torch.rand(N, generator=g)).to(dev)
    x = (torch.randn(M, N, generator=g) + 0.7).to(torch.bfloat16).to(dev)
    fn = {"none": lambda t: t, "gelu_erf": F.gelu, "gelu_tanh": lambda t: F.gelu(t, approximate="tanh"),
          "quick_gelu": lambda t: t * torch.sigmoid(1.702 * t)}[act]
    ref = x.float() + ls * fn(a.float() @ w.float().t() + bias)
    parts = torch.full((N // 64, M, 2), float("nan"), device=dev)  # column tile major: a wave's rows are one coalesced store
    out = x.clone()
    check(lib.merv_debug_gemm_stats(ptr(a), ptr(w), ptr(out), ptr(bias), ptr(ls), ptr(out), M, N, K, ACT[act], ptr(parts),
                                    torch.cuda.current_stream(dev).cuda_stream), "merv_debug_gemm_stats")
    assert rel_l2(out, ref) < 6e-3, (M, N, K, act)
    o = out.float()
    assert torch.isfinite(parts).all()
    parts = parts.transpose(0, 1)  # [M, N // 64, 2]
    mean = parts[..., 0].sum(1) / N
    m2 = (parts[..., 1] + 64 * (parts[..., 0] / 64 - mean[:, None]) ** 2).sum(1)
    assert torch.allclose(mean, o.mean(1), rtol=1e-4, atol=1e-5)
    assert torch.allclose(m2 / N, o.var(1, unbiased=False), rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("M,N,K,div,mod", [(2056, 1024, 256, 257, 8), (33000, 1024, 4096, 257, 8), (700, 256, 1024, 300, 3), (257 * 9 + 5, 1024, 1024, 257, 8)])
def test_epilogue_row_add(dev, M, N, K, div, mod):
    """GemmArgs::row_add (LanguageBind: fc2 adds the NEXT block's temporal embedding[frame of the row] behind its residual add, so the
    temporal LayerNorm folds into the temporal qkv GEMM): C = bf16(bf16(lin + res) + table[(m / div) % mod]) bit for bit against the
    two bf16 steps in torch, for every tile configuration the launcher picks (eight-phase part + remaining rows at 33000), and the
    LayerNorm partials describe the FINAL values."""
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * K**-0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    x = (torch.randn(M, N, generator=g) + 0.3).to(torch.bfloat16).to(dev)
    table = torch.randn(mod, N, generator=g).to(dev)
    parts = torch.full((N // 64, M, 2), float("nan"), device=dev)
    out = x.clone()
    check(lib.merv_debug_gemm_row_add(ptr(a), ptr(w), ptr(out), ptr(bias), ptr(out), M, N, K, ptr(table), div, mod, ptr(parts),
                                      torch.cuda.current_stream(dev).cuda_stream), "merv_debug_gemm_row_add")
    plain = x.clone()  # the same launch without the table: its output is bf16(lin + res)
    parts0 = torch.empty_like(parts)
    check(lib.merv_debug_gemm_stats(ptr(a), ptr(w), ptr(plain), ptr(bias), None, ptr(plain), M, N, K, 0, ptr(parts0),
                                    torch.cuda.current_stream(dev).cuda_stream), "merv_debug_gemm_stats")
    idx = (torch.arange(M, device=dev) // div) % mod
    ref = (plain.float() + table[idx]).to(torch.bfloat16)
    assert torch.equal(out, ref)
    o = out.float()
    parts = parts.transpose(0, 1)
    mean = parts[..., 0].sum(1) / N
    m2 = (parts[..., 1] + 64 * (parts[..., 0] / 64 - mean[:, None]) ** 2).sum(1)
    assert torch.allclose(mean, o.mean(1), rtol=1e-4, atol=1e-5)
    assert torch.allclose(m2 / N, o.var(1, unbiased=False), rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("variant", [7, 9])
def test_erf_gelu_epilogue_polynomial_against_exact_gelu(dev, variant):
    """The erf-GELU of the GEMM epilogue is a clamped degree-8 minimax polynomial of Phi (common.h activate2): push EVERY finite
    bf16 value in [-12, 12] through it (identity weight, so the accumulator is exactly x) and compare with torch's exact fp32 GELU:
    |error| <= 5e-5 absolute before the bf16 output rounding, i.e. at most one bf16 ulp of the exact result + 5e-5."""
    from merv_amd import _lib, ops
    lib = _lib.load()
    bits = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)
    vals = bits[torch.isfinite(bits.float()) & (bits.float().abs() <= 12.0)]
    K = N = 256
    M = (vals.numel() + K - 1) // K
    x = torch.zeros(M * K, dtype=torch.bfloat16)
    x[: vals.numel()] = vals
    # row m holds K values; out[m][n] = gelu(sum_k a[m][k] I[n][k]) = gelu(a[m][n])
    a = x.view(M, K).to(dev)
    w = torch.eye(N, K, dtype=torch.bfloat16, device=dev)
    lib.merv_debug_set_gemm_variant(variant)
    try:
        out = ops.gemm(a, w, act="gelu_erf")
    finally:
        lib.merv_debug_set_gemm_variant(0)
    xf = a.float().cpu()
    want = F.gelu(xf.double()).float()
    got = out.float().cpu()
    ulp = torch.maximum(want.abs() * 2.0**-8, torch.tensor(2.0**-133))  # bf16 spacing at the exact value's magnitude
    err = (got - want).abs()
    assert bool((err <= ulp + 5e-5).all()), float((err - ulp).max())
    assert float(err.max()) < 8e-3 and float(err[xf.abs() < 1.0].max()) < 2.1e-3
    assert bool((got[xf >= 4.25] == xf[xf >= 4.25]).all()) and bool((got[xf < -4.25] == 0).all())  # saturated exactly (at -4.25 itself Phi = 0.5 - Q(1) = 8e-6)
