"""GPU: every wired key of the Video Backbone Registry (merv/models/materialize.py:31-73) through its VideoBackbone class --
the token selections other than merv-full's four (languagebind/__init__.py:88-101, dinov2_video.py:140-151, vivit.py:106-118)
are slices / concatenations / means of the encoder's full token tensor (merv_encoder_forward_select, merv_mean_rows).
Checked against the oracle's select_tokens (itself pinned to the reference's forward bodies, tests/test_oracle_goldens.py) on
the oracle's hidden states of the same seeded encoder, reduced depth; tolerance as tests/test_encoder_gpu.py."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

CASES = [
    ("languagebind-video", "languagebind", None, 8), ("languagebind-video-averagetoken", "languagebind", "average", 8),
    ("languagebind-video-classemb", "languagebind", "classemb", 8), ("languagebind-video-noclass", "languagebind", "noclass", 8),
    ("languagebind-video-classemb-at-first", "languagebind", "classemb-at-first", 16),
    ("dinov2-video", "dinov2", "cls", 3), ("dinov2-video-all-tokens", "dinov2", "all-tokens", 3),
    ("dinov2-video-all-token-with-cls", "dinov2", "all-token-with-cls", 3), ("dinov2-video-classemb-at-first", "dinov2", "classemb-at-first", 3),
    ("vivit-google-b-cls-token", "vivit", "cls-token", 32), ("vivit-google-b-all-tokens", "vivit", "all-tokens", 32),
    ("vivit-google-b-all-no-cls", "vivit", "all-no-cls", 32), ("vivit-google-b-all-no-cls-16frames", "vivit", "all-no-cls-16frames", 32),
    ("vivit-google-b-classemb-at-first-16frames", "vivit", "classemb-at-first-16frames", 32),
]


@pytest.mark.parametrize("ident,family,rule,frames", CASES)
def test_registry_key_forward_equals_oracle_selection(dev, ident, family, rule, frames):
    from oracle import merv_oracle as O
    from oracle.parity import spec_to_cfg
    from merv_amd.backbones import VIDEO_BACKBONES, random_weights
    entry = VIDEO_BACKBONES[ident]
    # two blocks (the bare dinov2 id then applies the final norm to them; the others read them as the second-to-last output)
    bb = entry["cls"](ident, "resize-naive", num_frames=frames, weights="random", device=dev, layers=2, **entry["kwargs"])
    spec = bb.spec
    W = random_weights(spec, seed=spec.dim + spec.frames)  # what weights="random" built the encoder from
    cfg = spec_to_cfg(spec)
    B = 2
    pix = torch.randn(spec.pixel_shape(B), generator=torch.Generator().manual_seed(5))
    hidden = O.encoder_hidden(pix, cfg, W)
    ref = O.select_tokens(hidden, family, B, rule)
    out = bb(pix.to(dev), None)
    torch.cuda.synchronize()
    assert out.shape == ref.shape, (ident, out.shape, ref.shape)
    assert out.dtype == torch.bfloat16 and torch.isfinite(out.float()).all()
    err = rel_l2(out, ref)
    assert err < 2e-2, (ident, err)
    # the reference's own num_patches bookkeeping (its quirks included: a leading class token is not counted)
    extra = 1 if "classemb-at-first" in ident and family != "vivit" else 0
    if ident == "dinov2-video-all-token-with-cls":
        assert bb.num_patches == frames and out.shape[1] == frames * 257
    elif ident == "vivit-google-b-classemb-at-first-16frames":
        assert bb.num_patches == 3136 and out.shape[1] == 3137
    else:
        assert out.shape[1] == bb.num_patches + extra, (ident, out.shape, bb.num_patches)


def test_mean_rows_is_torch_mean_of_a_bf16_tensor(dev):
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    for groups, rows, D in [(32, 257, 1024), (2, 16, 1024), (5, 1, 768), (3, 7, 8)]:
        x = (torch.randn(groups, rows, D, generator=g) * 3).to(torch.bfloat16).to(dev)
        out = torch.empty(groups, D, dtype=torch.bfloat16, device=dev)
        check(lib.merv_mean_rows(ptr(x), ptr(out), groups, rows, D, rows, torch.cuda.current_stream(dev).cuda_stream), "mean")
        ref = x.float().mean(1)
        assert (out.float() - ref).abs().max() <= ref.abs().max() * 2**-8
        assert rel_l2(out, ref) < 3e-3


@pytest.mark.parametrize("ident", ["siglip-vit-b16-224px", "siglip-vit-b16-224px-all-tokens"])
def test_pooled_siglip_ids(dev, ident):
    """The SigLIP ids that keep timm's forward(): blocks + final norm + attention-pooling head, one feature per frame
    (siglip.py:46-63,142-151), against oracle.map_pool (pinned on transformers' SigLIP pooling head) of the oracle's hidden states."""
    from oracle import merv_oracle as O
    from oracle.parity import spec_to_cfg
    from merv_amd.backbones import VIDEO_BACKBONES, random_weights
    entry = VIDEO_BACKBONES[ident]
    bb = entry["cls"](ident, "resize-naive", num_frames=3, weights="random", device=dev, layers=2, **entry["kwargs"])
    spec = bb.spec
    assert spec.final_ln and bb.num_patches == 3 and bb.spatial_resolution == 1
    W = random_weights(spec, seed=spec.dim + spec.frames)
    B = 2
    pix = torch.randn(spec.pixel_shape(B), generator=torch.Generator().manual_seed(6))
    hidden = O.encoder_hidden(pix, spec_to_cfg(spec), W)  # [B*3, 196, 768] after the final norm
    ref = O.map_pool(hidden, bb.pool_weights, heads=spec.heads).reshape(B, 3, spec.dim)
    out = bb(pix.to(dev), None)
    torch.cuda.synchronize()
    assert out.shape == ref.shape and out.dtype == torch.bfloat16
    assert rel_l2(out, ref) < 2e-2, rel_l2(out, ref)


def test_map_pool_attention_kernel(dev):
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    for nseq, S, heads in [(6, 196, 12), (1, 1, 2), (3, 1000, 4)]:
        D = heads * 64
        kv = torch.randn(nseq * S, 2 * D, generator=g).to(torch.bfloat16).to(dev)
        q = (torch.randn(D, generator=g) * 2).to(dev)
        out = torch.empty(nseq, D, dtype=torch.bfloat16, device=dev)
        check(lib.merv_map_pool_attention(ptr(kv), ptr(q), ptr(out), nseq, S, heads, 0.125, torch.cuda.current_stream(dev).cuda_stream), "map")
        k = kv[:, :D].float().view(nseq, S, heads, 64)
        v = kv[:, D:].float().view(nseq, S, heads, 64)
        p = torch.softmax(torch.einsum("nshd,hd->nhs", k, q.view(heads, 64)) * 0.125, -1)
        ref = torch.einsum("nhs,nshd->nhd", p, v).reshape(nseq, D)
        assert rel_l2(out, ref) < 4e-3, (nseq, S, heads, rel_l2(out, ref))
