"""Rows a1 / a2 on the box that serves the product: the bit-exact index tests of tests/test_host_cpu.py again under the
`gpu` marker (they need no GPU -- the sampler is host code of libmerv_hip.so -- but the driver's `-m gpu` run is the one
that records which native library the product loads), plus the reference's own eval_data/dummy_mcq clip
(`end_frame=595`, /root/reference/eval_data/dummy_mcq/test_q.json) through merv_amd.video_io.load_video on a pre-decoded
clip, and the per-encoder stride of MERV.generate (merv.py:803-806) applied to what load_video returned."""
import json
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def test_sampler_bit_exact_through_c_abi_on_gpu_box():
    from merv_amd.sampler import frame_indices
    cases = json.loads((G / "frame_indices.json").read_text())
    assert len(cases) >= 80
    for c in cases:
        got = frame_indices(c["N"], c["fps"], c["clip_start_sec"], c["clip_end_sec"], c["num_frames"], c["end_frame"])
        assert got == c["ids"], c


def test_temporal_subsample_on_gpu_box():
    from merv_amd.sampler import temporal_subsample
    assert temporal_subsample(32, 32, 16) == list(range(0, 32, 2))
    assert temporal_subsample(32, 32, 32) == list(range(32))
    assert len(temporal_subsample(32, 32, 12)) == 16  # reference over-sampling quirk kept
    with pytest.raises(ValueError):
        temporal_subsample(32, 16, 32)


def test_dummy_mcq_end_frame_through_load_video(dev):
    """The reference's dummy MCQ sample: 596 decoded frames at 29.97 fps, end_frame=595, 32 frames loaded, then strides
    [2, 2, 1, 2]. Frame n of the synthetic clip carries n in its pixels, so the frames that come back name their index."""
    from merv_amd.sampler import temporal_subsample
    from merv_amd.video_io import load_video
    case = next(c for c in json.loads((G / "frame_indices.json").read_text()) if c["end_frame"] == 595 and c["N"] == 596)
    N = case["N"]
    clip = torch.zeros(N, 4, 6, 3, dtype=torch.uint8)
    clip[:, 0, 0, 0] = torch.arange(N) % 256
    clip[:, 0, 0, 1] = torch.arange(N) // 256
    frames = load_video((clip, case["fps"]), num_frames=32, end_frame=595).to(dev)
    assert frames.shape == (32, 3, 4, 6) and frames.dtype == torch.uint8
    got = (frames[:, 0, 0, 0].long() + 256 * frames[:, 1, 0, 0].long()).tolist()
    assert got == case["ids"]
    for nf, stride in zip([16, 16, 32, 16], [2, 2, 1, 2]):
        idx = temporal_subsample(frames.shape[0], 32, nf)
        assert idx == list(range(0, 32, stride))
        sub = frames[idx]
        assert (sub[:, 0, 0, 0].long() + 256 * sub[:, 1, 0, 0].long()).tolist() == case["ids"][::stride]
