"""GPU parity tests of the single HIP kernels (through the C ABI) against plain fp32 torch references computed from
the SAME bf16-rounded inputs. Tolerances: outputs are bf16 (8 significand bits => 2^-9 relative rounding), fp32
accumulation; rel-L2 <= 6e-3 everywhere, elementwise |err| <= 2^-7 * max|ref| (stated per test)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL = 6e-3


def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1, 128, 64), (257, 256, 640), (4112, 1024, 1024),
                                   (3137, 768, 3072), (1024, 4096, 768), (130, 2304, 768)])
def test_gemm_plain(dev, M, N, K):
    from merv_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * K**-0.5).to(dev)
    out = ops.gemm(a, w)
    ref = a.float() @ w.float().t()
    assert rel_l2(out, ref) < TOL
    assert (out.float() - ref).abs().max() <= 2**-7 * ref.abs().max()


def test_gemm_asymmetric_identity(dev):
    """A = I with an asymmetric W catches a transposed C/D map (cdna_hip_programming.md section 3)."""
    from merv_amd import ops
    K = N = 128
    a = _bf(torch.eye(K)).to(dev)
    w = _bf(torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125).to(dev)
    out = ops.gemm(a, w)
    assert torch.equal(out.float(), w.float().t())


@pytest.mark.parametrize("act", ["none", "gelu_erf", "gelu_tanh", "quick_gelu"])
def test_gemm_epilogue(dev, act):
    from merv_amd import ops
    M, N, K = 523, 384, 256
    g = torch.Generator().manual_seed(3)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * K**-0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ls = (0.5 + torch.rand(N, generator=g)).to(dev)
    res = _bf(torch.randn(M, N, generator=g)).to(dev)
    out = ops.gemm(a, w, bias=bias, act=act, lscale=ls, res=res)
    y = a.float() @ w.float().t() + bias
    if act == "gelu_erf":
        y = F.gelu(y)
    elif act == "gelu_tanh":
        y = 0.5 * y * (1 + torch.tanh(y * 0.7978845608 * (1 + 0.044715 * y * y)))
    elif act == "quick_gelu":
        y = y * torch.sigmoid(1.702 * y)
    ref = res.float() + ls * y
    assert rel_l2(out, ref) < TOL


def test_gemm_inplace_residual(dev):
    from merv_amd import ops
    M, N, K = 300, 256, 128
    g = torch.Generator().manual_seed(5)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * K**-0.5).to(dev)
    x = _bf(torch.randn(M, N, generator=g)).to(dev)
    ref = x.float() + a.float() @ w.float().t()
    ops.gemm(a, w, res=x, out=x)
    assert rel_l2(x, ref) < TOL


def test_gemm_embed_scatter(dev):
    """Patch-embedding epilogue: position rows m % P, output rows scattered past the prefix tokens."""
    from merv_amd import ops
    nseq, P, pre, N, K = 3, 49, 5, 128, 64
    g = torch.Generator().manual_seed(9)
    a = _bf(torch.randn(nseq * P, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * K**-0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    pos = _bf(torch.randn(P, N, generator=g)).to(dev)
    out = torch.zeros(nseq * (P + pre), N, dtype=torch.bfloat16, device=dev)
    ops.gemm(a, w, bias=bias, res=pos, res_row_mod=P, out=out, out_group=P, out_stride=P + pre, out_off=pre)
    ref = (a.float() @ w.float().t() + bias).reshape(nseq, P, N) + pos.float()[None]
    got = out.reshape(nseq, P + pre, N)
    assert rel_l2(got[:, pre:], ref) < TOL
    assert torch.count_nonzero(got[:, :pre]) == 0


def test_gemm_rejects_bad_shapes(dev):
    from merv_amd import ops
    a = torch.zeros(8, 100, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(128, 100, dtype=torch.bfloat16, device=dev)
    with pytest.raises(ValueError):
        ops.gemm(a, w)


@pytest.mark.parametrize("M,D", [(1, 768), (7, 1024), (4112, 1024), (3137, 768)])
def test_layernorm(dev, M, D):
    from merv_amd import ops
    g = torch.Generator().manual_seed(M + D)
    x = _bf(torch.randn(M, D, generator=g) * 3 + 1.5).to(dev)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    beta = (0.1 * torch.randn(D, generator=g)).to(dev)
    y = ops.layernorm(x, gamma, beta, 1e-6)
    ref = F.layer_norm(x.float(), (D,), gamma, beta, 1e-6)
    assert rel_l2(y, ref) < TOL


def test_layernorm_temporal_add(dev):
    from merv_amd import ops
    ntok, t, nclip, D = 13, 8, 2, 1024
    M = nclip * t * ntok
    g = torch.Generator().manual_seed(1)
    x = _bf(torch.randn(M, D, generator=g)).to(dev)
    temb = (torch.randn(t, D, generator=g) * 0.5).to(dev)
    gamma = torch.ones(D, device=dev)
    beta = torch.zeros(D, device=dev)
    x0 = x.clone()
    y = ops.layernorm(x, gamma, beta, 1e-5, add=temb, add_div=ntok, add_mod=t)
    frame = (torch.arange(M, device=dev) // ntok) % t
    xr = _bf(x0.float() + temb[frame])
    assert torch.equal(x, xr)  # residual stream updated in place, rounded once
    assert rel_l2(y, F.layer_norm(xr.float(), (D,), gamma, beta, 1e-5)) < TOL


def _attn_ref(qkv, nseq, L, heads):
    D = heads * 64
    q, k, v = qkv.float().reshape(nseq, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    att = (q @ k.transpose(-1, -2)) * 0.125
    return (att.softmax(-1) @ v).transpose(1, 2).reshape(nseq * L, D)


@pytest.mark.parametrize("vtr", ["1", "0"])
@pytest.mark.parametrize("nseq,L,heads", [(2, 64, 1), (3, 196, 12), (2, 257, 16), (2, 261, 16), (1, 3137, 12), (1, 1, 2),
                                          (1, 65, 2)])
def test_attention(dev, hooks_library, vtr, nseq, L, heads):
    from merv_amd import ops
    os.environ["MERV_ATTN_VTR"] = vtr
    try:
        g = torch.Generator().manual_seed(L)
        qkv = _bf(torch.randn(nseq * L, 3 * heads * 64, generator=g) * 1.5).to(dev)
        out = ops.attention(qkv, nseq, L, heads)
        ref = _attn_ref(qkv, nseq, L, heads)
        assert rel_l2(out, ref) < 1e-2, (vtr, nseq, L, heads)
    finally:
        os.environ.pop("MERV_ATTN_VTR", None)


def test_attention_spiked_max(dev):
    """Force the online-softmax rescale: one key in a late tile dominates one query (rule 26)."""
    from merv_amd import ops
    L, heads = 300, 2
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(L, 3 * heads * 64, generator=g)
    qkv[7, :64] = 6.0           # query 7, head 0
    qkv[290, 128:192] = 6.0     # key 290, head 0 -> score 6*6*64/8 = 288
    qkv = _bf(qkv).to(dev)
    out = ops.attention(qkv, 1, L, heads)
    ref = _attn_ref(qkv, 1, L, heads)
    assert rel_l2(out, ref) < 1e-2
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("L", [196, 257, 300, 3137])
def test_attention_first_tile_reference(dev, hooks_library, L):
    """Extreme first key tiles: queries whose scores against EVERY key are ~ -128 nats, ~ +128 nats, and one whose first tile is
    ~ -128 while a later key is ~ +40 (the running reference has to start from the tile's own maximum and move far); each at the
    shipped threshold, thr = 0 and thr = 64."""
    from merv_amd import _lib, ops
    lib = _lib.load()
    heads, D = 2, 128
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(L, 3 * D, generator=g) * 0.3
    qkv[:, D:D + 64] += 4.0                 # every key of head 0 points along +1 (norm ~ 4 * 8)
    qkv[5, :64] = -4.0                      # query 5: scores ~ -4 * 4 * 64 / 8 = -128 nats against all keys
    qkv[6, :64] = 4.0                       # query 6: ~ +128 nats against all keys
    qkv[40, :64] = -4.0                     # query 40: ~ -128 everywhere ...
    qkv[L - 3, D:D + 64] = -1.25            # ... except one late key: +40 nats
    qkv = _bf(qkv).to(dev)
    ref = _attn_ref(qkv, 1, L, heads)
    try:
        for thr in (8.0, 0.0, 64.0):
            lib.merv_debug_set_attn_rescale_thr(thr)
            out = ops.attention(qkv, 1, L, heads)
            assert torch.isfinite(out.float()).all(), thr
            assert rel_l2(out, ref) < 1e-2, (thr, L)
            per_row = ((out.float().cpu() - ref.cpu()).norm(dim=-1) / (ref.cpu().norm(dim=-1) + 1e-20))
            assert float(per_row.max()) < 3e-2, (thr, L, int(per_row.argmax()))
    finally:
        lib.merv_debug_set_attn_rescale_thr(8.0)


@pytest.mark.parametrize("vtr", ["1", "0"])
@pytest.mark.parametrize("L", [257, 258, 261, 264, 265])
def test_attention_extra_rows_split_over_key_tiles(dev, hooks_library, vtr, L):
    """257 / 261-token sequences run 8 full query tiles on a 4 x 2 block; the 1..8 rows past them are multiplied against one
    key tile per wave and merged through LDS (attention.hip, XQ). A whole-tensor norm would hide one wrong row in 257, so
    the extra rows are checked on their own -- with a score spike in an early AND a late key tile for the first extra row, so
    the partials being merged carry very different maxima (265 rows fall back to the 3 x 3 block: same checks)."""
    from merv_amd import ops
    os.environ["MERV_ATTN_VTR"] = vtr
    try:
        nseq, heads = 3, 4
        g = torch.Generator().manual_seed(L)
        qkv = torch.randn(nseq * L, 3 * heads * 64, generator=g) * 1.2
        D = heads * 64
        qkv[L + 256, 64:128] = 3.0                 # sequence 1, head 1: query row 256 ...
        qkv[L + 5, D + 64:D + 128] = 3.0           # ... loves key 5 (tile 0): score 3*3*64/8 = 72
        qkv[L + 200, D + 64:D + 128] = 2.9         # ... and key 200 (tile 3) almost as much
        qkv = _bf(qkv).to(dev)
        out = ops.attention(qkv, nseq, L, heads).float().cpu().reshape(nseq, L, D)
        ref = _attn_ref(qkv, nseq, L, heads).float().cpu().reshape(nseq, L, D)
        assert torch.isfinite(out).all()
        assert rel_l2(out, ref) < 1e-2
        assert rel_l2(out[:, 256:], ref[:, 256:]) < 1e-2, (vtr, L)
        assert rel_l2(out[1, 256, 64:128], ref[1, 256, 64:128]) < 1e-2
        per_row = ((out - ref).norm(dim=-1) / (ref.norm(dim=-1) + 1e-20))
        assert float(per_row.max()) < 3e-2, (vtr, L, int(per_row.argmax()))
    finally:
        os.environ.pop("MERV_ATTN_VTR", None)


@pytest.mark.parametrize("vtr", ["1", "0"])
@pytest.mark.parametrize("nclips,ntok,heads", [(1, 4, 1), (2, 257, 16), (1, 3, 2), (3, 17, 4)])
def test_temporal_attention(dev, hooks_library, vtr, nclips, ntok, heads):
    from merv_amd import ops
    os.environ["MERV_ATTN_VTR"] = vtr
    try:
        t, D = 8, heads * 64
        g = torch.Generator().manual_seed(ntok)
        qkv = _bf(torch.randn(nclips * t * ntok, 3 * D, generator=g) * 1.5).to(dev)
        out = ops.temporal_attention(qkv, nclips, t, ntok, heads)
        # "(b t) n d -> (b n) t d" then plain attention over t (modeling_video.py:145-155)
        x = qkv.reshape(nclips, t, ntok, 3 * D).permute(0, 2, 1, 3).reshape(nclips * ntok * t, 3 * D)
        ref = _attn_ref(x, nclips * ntok, t, heads).reshape(nclips, ntok, t, D).permute(0, 2, 1, 3).reshape(-1, D)
        assert rel_l2(out, ref) < 1e-2
    finally:
        os.environ.pop("MERV_ATTN_VTR", None)


@pytest.mark.parametrize("layout,patch,tub,frames,dtype", [("BFCHW", 16, 1, 2, torch.float32), ("BCFHW", 14, 1, 3, torch.float32),
                                                           ("BFCHW", 16, 2, 4, torch.bfloat16), ("BFCHW", 14, 1, 3, torch.bfloat16),
                                                           ("BCFHW", 16, 1, 2, torch.bfloat16), ("BFCHW", 16, 2, 4, torch.float32),
                                                           ("BFCHW", 32, 1, 2, torch.bfloat16), ("BFCHW", 8, 2, 2, torch.float32)])
def test_im2col(dev, layout, patch, tub, frames, dtype):
    from merv_amd import ops
    B, img = 2, 224
    g = torch.Generator().manual_seed(2)
    shape = (B, frames, 3, img, img) if layout == "BFCHW" else (B, 3, frames, img, img)
    pix = torch.randn(shape, generator=g).to(dtype).to(dev)
    ktrue = 3 * tub * patch * patch
    kpad = (ktrue + 63) // 64 * 64
    col = ops.im2col(pix, layout, patch, tub, kpad)
    x = pix.float() if layout == "BFCHW" else pix.float().permute(0, 2, 1, 3, 4)
    hp = img // patch
    # [B, F/t, t, 3, hp, p, hp, p] -> [B, F/t, hp, hp, 3, t, p, p]
    x = x.reshape(B, frames // tub, tub, 3, hp, patch, hp, patch).permute(0, 1, 4, 6, 3, 2, 5, 7)
    ref = x.reshape(B * (frames // tub) * hp * hp, ktrue)
    assert torch.equal(col[:, :ktrue].float(), _bf(ref).float())
    assert torch.count_nonzero(col[:, ktrue:]) == 0


@pytest.mark.parametrize("S,C", [(16, 1024), (14, 768)])
def test_pool3d(dev, S, C):
    from merv_amd import ops
    B, T, O = 2, 16, 8
    g = torch.Generator().manual_seed(S)
    tok = _bf(torch.randn(B, T * S * S, C, generator=g)).to(dev)
    out = ops.pool3d(tok, T, S, O)
    x = tok.float().reshape(B, T, S, S, C).permute(0, 4, 1, 2, 3)
    ref = F.adaptive_avg_pool3d(x, (T, O, O)).permute(0, 2, 3, 4, 1).reshape(B, T * O * O, C)
    assert rel_l2(out, ref) < TOL


def test_fusion_and_splice(dev):
    from merv_amd.projector import CrossAttentionAdapterLearnableQuery, splice
    B, E, T, C, Ed = 2, 4, 1024, 512, 384
    torch.manual_seed(0)
    fus = CrossAttentionAdapterLearnableQuery(embed_dim=Ed, llm_dim=C, token_length=T, averagetoken=True)
    with torch.no_grad():
        fus.attention.in_proj_bias.normal_(0, 0.05)
        fus.Q.mul_(20)
    V = [_bf(torch.randn(B, T, C) + 0.3 * e).to(dev) for e in range(E)]
    out, w = fus(V)
    # literal reference order of operations (nn_utils.py:500-521) in fp32
    Vs = torch.stack([v.float().cpu() for v in V], 1)
    p, wref = fus.attention(query=fus.Q.repeat(B, 1).unsqueeze(1), key=Vs.mean(2), value=Vs.mean(2))
    ref = torch.bmm(wref, Vs.reshape(B, E, T * C)).reshape(B, T, C)
    assert (w.cpu() - wref[:, 0]).abs().max() < 2e-3
    assert abs(float(w.sum()) - B) < 1e-4
    assert rel_l2(out, ref.detach()) < TOL
    emb = _bf(torch.randn(B, 9, C)).to(dev)
    sp = splice(emb, out, 1)
    assert torch.equal(sp, torch.cat([emb[:, :1], out, emb[:, 1:]], 1))
    sp0 = splice(emb, out, 0)
    assert torch.equal(sp0, torch.cat([out, emb], 1))


@pytest.mark.parametrize("L,late_key", [(300, 290), (3137, 3000), (261, 250)])
def test_attention_deferred_max(dev, hooks_library, L, late_key):
    """The online softmax moves a query's reference only when a key tile's maximum exceeds it by more than 2^thr (attention.hip,
    deferred max; rule 26 of the guide: force the branch, sweep the threshold). Three queries of head 0 see late keys whose
    scores jump by (a) less than the threshold -- exponentials > 1 against the kept reference --, (b) just above it, (c) far
    above it; the shipped threshold, the exact running maximum (thr = 0) and "never but overflow" (thr = 64) must all match
    the fp32 reference, and each other to rounding."""
    from merv_amd import ops
    heads, D = 2, 128
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(L, 3 * D, generator=g) * 0.5
    # query rows 3, 40, 70 (head 0) all point along +1; late keys of different lengths give score jumps of ~3.5, ~7 and ~40 nats
    for row in (3, 40, 70):
        qkv[row, :64] = 1.0
    qkv[late_key, D:D + 64] = 0.45        # score 0.45 * 64 / 8 = 3.6   (5.2 binary orders: below thr = 8)
    qkv[late_key + 1, D:D + 64] = 0.9     # 7.2 nats = 10.4 binary orders: above
    qkv[late_key + 2, D:D + 64] = 5.0     # 40 nats
    qkv = _bf(qkv).to(dev)
    ref = _attn_ref(qkv, 1, L, heads)
    outs = {}
    from merv_amd import _lib
    lib = _lib.load()
    try:
        for thr in ("8", "0", "64"):
            lib.merv_debug_set_attn_rescale_thr(float(thr))
            out = ops.attention(qkv, 1, L, heads)
            assert torch.isfinite(out.float()).all(), thr
            assert rel_l2(out, ref) < 1e-2, (thr, L)
            per_row = ((out.float().cpu() - ref.cpu()).norm(dim=-1) / (ref.cpu().norm(dim=-1) + 1e-20))
            assert float(per_row.max()) < 3e-2, (thr, L, int(per_row.argmax()))
            outs[thr] = out.float().cpu()
    finally:
        lib.merv_debug_set_attn_rescale_thr(8.0)
    assert rel_l2(outs["8"], outs["0"]) < 6e-3 and rel_l2(outs["64"], outs["0"]) < 6e-3
