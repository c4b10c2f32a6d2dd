"""GPU: the reference-shaped front door (registry -> VideoBackbone classes -> MERVVisual) against the oracle.
Config 1 of BASELINE.json (DINOv2-only single-encoder model, 4 frames) and a reduced-depth merv-full."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _oracle_cfg(spec):
    from oracle import merv_oracle as O
    return O.EncoderCfg(**{k: getattr(spec, k) for k in O.EncoderCfg.__dataclass_fields__})


def test_config1_dinov2_single_4_frames(dev):
    from oracle import merv_oracle as O
    from merv_amd.backbones import VIDEO_BACKBONES
    from merv_amd.vidlm import MERVVisual
    cfg = VIDEO_BACKBONES["dinov2-video-all-tokens"]
    bb = cfg["cls"]("dinov2-video-all-tokens", "resize-naive", num_frames=4, weights="random", device=dev, layers=2,
                    **cfg["kwargs"])
    bbs = [bb]
    assert (bb.embed_dim, bb.num_patches, bb.spatial_resolution, bb.temporal_resolution) == (1024, 4 * 256, 256, 4)
    assert bb.default_video_resolution == (4, 3, 224, 224)
    m = MERVVisual(bbs, llm_dim=512, visual_feature_length=512)  # single encoder: length is corrected to 4*64 (merv.py:194-205)
    assert m.visual_feature_length == 256
    g = torch.Generator().manual_seed(0)
    pix = torch.randn(1, 4, 3, 224, 224, generator=g)
    emb = torch.randn(1, 7, 512, generator=g).to(torch.bfloat16)
    am = torch.ones(1, 7, dtype=torch.bool)
    out, mask, _, w = m.forward_visual([pix.to(dev)], emb.to(dev), am.to(dev))
    assert out.shape == (1, 7 + 256, 512) and mask.shape == (1, 263) and bool(mask.all())
    assert torch.allclose(w.cpu(), torch.ones(1, 1))  # one encoder: fusion weight == 1, fused == projected
    # oracle with the very same parameters
    from merv_amd.backbones import random_weights
    Wc = random_weights(bb.spec, seed=bb.spec.dim + bb.spec.frames)
    tok = O.encoder_forward(pix, _oracle_cfg(bb.spec), Wc)
    lin = m.projectors[0].projector.projector
    ref = O.projector_forward(tok, 4, 16, 8, lin.weight.detach(), lin.bias.detach())
    assert rel_l2(out[:, 1:257], ref) < 2e-2
    assert torch.equal(out[:, :1].cpu(), emb[:, :1]) and torch.equal(out[:, 257:].cpu(), emb[:, 1:])


def test_merv_full_reduced_depth_through_registry(dev):
    from oracle import merv_oracle as O
    from merv_amd.backbones import VIDEO_BACKBONES, random_weights
    from merv_amd.vidlm import MERVVisual
    ids = ["languagebind-video-noclass", "dinov2-video-all-tokens", "vivit-google-b-all-no-cls-16frames",
           "siglip-vit-b16-224px-all-no-cls"]  # merv/conf/models.py:106-113
    frames = [16, 16, 32, 16]
    bbs = [VIDEO_BACKBONES[i]["cls"](i, "resize-naive", num_frames=f, weights="random", device=dev, layers=1,
                                     **VIDEO_BACKBONES[i]["kwargs"]) for i, f in zip(ids, frames)]
    m = MERVVisual(bbs, llm_dim=1024, visual_feature_length=1024)
    with torch.no_grad():
        m.feature_fusion.Q.mul_(30)
    g = torch.Generator().manual_seed(1)
    pix = [torch.randn(b.default_video_resolution, generator=g)[None] for b in bbs]
    fused, w = m.encode([p.to(dev) for p in pix])
    torch.cuda.synchronize()
    projected = []
    for b, p, pr in zip(bbs, pix, m.projectors):
        Wc = random_weights(b.spec, seed=b.spec.dim + b.spec.frames)
        tok = O.encoder_forward(p, _oracle_cfg(b.spec), Wc)
        lin = pr.projector.projector
        projected.append(O.projector_forward(tok, 16, b.spec.hp, 8, lin.weight.detach(), lin.bias.detach()))
    Fw = {k: v.detach() for k, v in m.feature_fusion.state_dict().items()}
    ref, wref = O.fusion_forward(projected, Fw)
    assert fused.shape == (1, 1024, 1024)
    assert (w.cpu() - wref).abs().max() < 5e-3
    assert rel_l2(fused, ref) < 2e-2


def test_registry_token_selections_through_merv(dev):
    """Registry ids whose forward() is not the plain patch selection, THROUGH MERV (VERDICT r4 missing #4; materialize.py:31-73 consumed
    at merv.py:563-585): `languagebind-video-classemb` (the class token of every frame) and `languagebind-video-averagetoken` (the mean
    over a frame's 257 tokens) give one token per frame, which the reference reshapes to [B, T, 1, C] and AveragePooling3DProjector
    pools UP to (T, 8, 8) (every output cell = that token) before the Linear; fused with a plain patch-token DINOv2. Against the
    oracle's select_tokens + projector_forward + fusion_forward on the same parameters."""
    from oracle import merv_oracle as O
    from merv_amd.backbones import VIDEO_BACKBONES, random_weights
    from merv_amd.vidlm import MERVVisual
    ids = ["languagebind-video-classemb", "languagebind-video-averagetoken", "dinov2-video-all-tokens"]
    rules = ["classemb", "average", None]
    bbs = [VIDEO_BACKBONES[i]["cls"](i, "resize-naive", num_frames=16, weights="random", device=dev, layers=2, **VIDEO_BACKBONES[i]["kwargs"])
           for i in ids]
    assert [b.selects_spec_patches for b in bbs] == [False, False, True]
    assert [(b.num_patches, b.spatial_resolution, b.temporal_resolution) for b in bbs] == [(16, 1, 16), (16, 1, 16), (4096, 256, 16)]
    m = MERVVisual(bbs, llm_dim=1024, visual_feature_length=1024)
    with torch.no_grad():
        m.feature_fusion.Q.mul_(30)
    g = torch.Generator().manual_seed(3)
    pix = [torch.randn(b.default_video_resolution, generator=g)[None].repeat(2, 1, 1, 1, 1) * (1.0 + 0.1 * k) for k, b in enumerate(bbs)]
    pix = [p + 0.05 * torch.randn(p.shape, generator=g) for p in pix]  # two different videos
    fused, w = m.encode([p.to(dev) for p in pix])
    fused, w = fused.clone(), w.clone()
    torch.cuda.synchronize()
    again, w2 = m.encode([p.to(dev) for p in pix])  # persistent buffers, side streams: the second call reproduces the first
    assert torch.equal(again, fused) and torch.equal(w2, w)
    projected = []
    for b, p, pr, rule in zip(bbs, pix, m.projectors, rules):
        Wc = random_weights(b.spec, seed=b.spec.dim + b.spec.frames)
        lin = pr.projector.projector
        if rule is None:
            tok = O.encoder_forward(p, _oracle_cfg(b.spec), Wc)
            projected.append(O.projector_forward(tok, 16, 16, 8, lin.weight.detach(), lin.bias.detach()))
        else:
            hidden = O.encoder_hidden(p, _oracle_cfg(b.spec), Wc)
            tok = O.select_tokens(hidden, "languagebind", 2, rule)
            assert tok.shape == (2, 16, 1024)
            projected.append(O.projector_forward(tok, 16, 1, 8, lin.weight.detach(), lin.bias.detach()))
    Fw = {k: v.detach() for k, v in m.feature_fusion.state_dict().items()}
    ref, wref = O.fusion_forward(projected, Fw)
    assert fused.shape == (2, 1024, 1024)
    assert (w.cpu() - wref).abs().max() < 5e-3
    assert rel_l2(fused, ref) < 2e-2
    # the per-encoder projected tokens of the class-token branch: 64 identical rows per frame (the 1 x 1 grid pooled up to 8 x 8)
    proj0 = m.visual_path(dev).buffers(0, 2)["proj"].view(2, 16, 64, 1024)
    assert torch.equal(proj0, proj0[:, :, :1].expand_as(proj0))
    with pytest.raises(NotImplementedError):
        m.visual_path(dev).encode_tokens([p.to(dev) for p in pix])  # the training step drives patch selections only


def test_hipgraph_replay_matches_eager(dev):
    """MervVisualPath.capture: the replayed graph (four encoder chains forked onto side streams inside the capture) gives
    bit-identical fused tokens, and follows new pixel values copied into its static inputs."""
    import bench
    from merv_amd.encoder import merv_full_specs
    import dataclasses
    from merv_amd.backbones import random_weights
    from merv_amd.projector import CrossAttentionAdapterLearnableQuery
    from merv_amd.visual_path import MervVisualPath
    specs = [dataclasses.replace(s, layers=1) for s in merv_full_specs()]
    torch.manual_seed(0)
    enc_w = [random_weights(s, seed=i) for i, s in enumerate(specs)]
    proj_w = [(torch.randn(256, s.dim) * 0.02, torch.zeros(256)) for s in specs]
    fusion = CrossAttentionAdapterLearnableQuery(embed_dim=96, llm_dim=256, token_length=1024, averagetoken=True)
    path = MervVisualPath(specs, enc_w, proj_w, fusion, dev)
    g = torch.Generator().manual_seed(3)
    pix = [torch.randn(s.pixel_shape(1), generator=g).to(dev) for s in specs]
    pix2 = [torch.randn(s.pixel_shape(1), generator=g).to(dev) for s in specs]
    eager, w = path.forward(pix)
    eager, w = eager.clone(), w.clone()
    eager2 = path.forward(pix2)[0].clone()
    replay = path.capture(pix)
    f, wr = replay()
    torch.cuda.synchronize()
    assert torch.equal(f, eager) and torch.equal(wr, w)
    f2, _ = replay(pix2)
    torch.cuda.synchronize()
    assert torch.equal(f2, eager2) and not torch.equal(eager2, eager)
