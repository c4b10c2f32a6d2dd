"""GPU: BASELINE.json configs[1] at the depth and width the bench runs -- ONE video through the four encoders
(23 / 23 / 12 / 11 consumed blocks), the projectors and the fusion, HIP path vs the fp32 CPU oracle ON THE SAME WEIGHTS
AND PIXELS (merv/models/vidlms/merv.py:562-609). Asserts the stated bf16 tolerance where it matters: per-encoder tokens,
projected tokens and the fused [1,1024,4096] output, rel-L2 <= 2e-2 and min per-token cosine >= 0.999.

Also: a frame-range forward (merv_encoder_forward_frames: the multi-GPU placement's unit) returns exactly the matching
rows of the whole-video forward, and MERV.encode() runs the very MervVisualPath the bench times."""
import dataclasses
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


@pytest.mark.parametrize("batch", [1, 8, 16])
def test_full_depth_full_width_parity_vs_oracle(dev, batch):
    """batch = 16 is the bench default (8 was, in rounds 1-2): the compared video is the last of the batch, whose rows are the ones every GEMM's
    second launch (the rows behind the round-filling split) computes."""
    import bench
    specs, _, path, extras = bench.build_models(dev, concurrent=True, want_ref=True)
    par, cpu = bench.parity_and_cpu_baseline(path, specs, extras["ref"], dev, batch=batch)
    print("full-depth parity:", par)
    print("oracle:", cpu["sample"])
    for name, v in par["encoders"].items():
        assert v["tokens"]["rel_l2"] <= bench.TOL_REL_L2, (name, v)
        assert v["tokens"]["min_cos"] >= bench.TOL_MIN_COS, (name, v)
        assert v["projected"]["rel_l2"] <= bench.TOL_REL_L2, (name, v)
    assert par["fused"]["rel_l2"] <= bench.TOL_REL_L2, par["fused"]
    assert par["fused"]["min_cos"] >= bench.TOL_MIN_COS, par["fused"]
    assert par["fusion_weights_max_abs_diff"] <= 2e-3
    assert par["pass"] and par["depth"] == "23/23/12/11"


@pytest.mark.parametrize("name,f0,f1", [("languagebind", 8, 16), ("dinov2", 3, 13), ("siglip", 0, 5)])
def test_frame_range_forward_equals_rows_of_whole_forward(dev, name, f0, f1):
    """Frames are independent sequences (LanguageBind: clips of 8), and the projector pools inside a frame: the unit
    (encoder, video, frames [f0, f1)) must reproduce rows [f0*64, f1*64) of the whole projection bit for bit."""
    from merv_amd.backbones import random_weights
    from merv_amd.encoder import merv_full_specs
    from merv_amd.projector import CrossAttentionAdapterLearnableQuery
    from merv_amd.visual_path import MervVisualPath
    spec = dataclasses.replace(next(s for s in merv_full_specs() if s.name == name), layers=2)
    W = random_weights(spec, seed=3, device=dev)
    g = torch.Generator(device=dev).manual_seed(9)
    pw = (torch.randn(4096, spec.dim, generator=g, device=dev) * spec.dim**-0.5, torch.randn(4096, generator=g, device=dev) * 0.02)
    path = MervVisualPath([spec], [W], [pw], None, dev)
    pix = torch.randn(spec.pixel_shape(2), generator=g, device=dev).to(torch.bfloat16)
    whole = path.encode_project(0, pix).clone()
    part_pix = (pix[:, :, f0:f1] if spec.pix_layout == "BCFHW" else pix[:, f0:f1]).contiguous()
    part = path.encode_project(0, part_pix, frames=f1 - f0)
    torch.cuda.synchronize()
    assert part.shape == (2, (f1 - f0) * 64, 4096)
    assert torch.equal(part, whole[:, f0 * 64:f1 * 64])


def test_frame_range_rejects_what_does_not_split(dev):
    from merv_amd.backbones import random_weights
    from merv_amd.encoder import HipEncoder, merv_full_specs
    specs = {s.name: dataclasses.replace(s, layers=1) for s in merv_full_specs()}
    viv = HipEncoder(specs["vivit"], random_weights(specs["vivit"], seed=1, device=dev), dev)
    with pytest.raises(ValueError):
        viv.forward(torch.zeros(1, 16, 3, 224, 224, device=dev), frames=16)  # joint space-time attention: whole videos only
    lb = HipEncoder(specs["languagebind"], random_weights(specs["languagebind"], seed=1, device=dev), dev)
    with pytest.raises(ValueError):
        lb.forward(torch.zeros(1, 3, 4, 224, 224, device=dev), frames=4)  # not a whole clip of 8


def test_generate_path_is_the_benched_path(dev):
    """MERV.encode() (the generate() front door's visual branch) delegates to MervVisualPath: same object type as the
    bench's, persistent buffers (two calls return the same storage), and results bit-equal to a path built by hand from
    the same encoders and parameters."""
    from merv_amd.backbones import VIDEO_BACKBONES
    from merv_amd.vidlm import MERVVisual
    from merv_amd.visual_path import MervVisualPath
    ids = ["dinov2-video-all-tokens", "siglip-vit-b16-224px-all-no-cls"]
    bbs = [VIDEO_BACKBONES[i]["cls"](i, "resize-naive", num_frames=4, weights="random", device=dev, layers=2, **VIDEO_BACKBONES[i]["kwargs"])
           for i in ids]
    m = MERVVisual(bbs, llm_dim=4096, visual_feature_length=256)
    g = torch.Generator(device=dev).manual_seed(0)
    pix = [torch.randn(1, 4, 3, 224, 224, generator=g, device=dev) for _ in ids]
    fused1, w1 = m.encode(pix)
    assert isinstance(m.visual_path(dev), MervVisualPath)
    keep = fused1.clone()
    fused2, w2 = m.encode(pix)
    assert fused2.data_ptr() == fused1.data_ptr() and torch.equal(fused2, keep)  # persistent, deterministic
    hand = MervVisualPath([b.spec for b in bbs], None,
                          [(p.projector.projector.weight, p.projector.projector.bias) for p in m.projectors], m.feature_fusion,
                          dev, encoders=[b.featurizer for b in bbs])
    f3, w3 = hand.forward(pix)
    torch.cuda.synchronize()
    assert torch.equal(f3, keep) and torch.equal(w3, w1)
    # parameters changed in place (an optimizer step, a checkpoint load) are picked up by the next encode()
    with torch.no_grad():
        m.projectors[0].projector.projector.bias.add_(1.0)
    fused4, _ = m.encode(pix)
    assert not torch.equal(fused4, keep)
