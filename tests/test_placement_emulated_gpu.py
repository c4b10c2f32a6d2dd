"""GPU: the multi-GPU placement (merv_amd/distributed.py) with the REAL kernels, every rank of a world of
2 / 4 / 8 played in turn on this one device. Each emulated rank runs its own units (encoder x videos x frame range) through
encoder + projector and packs its rows (`produce`); the collective is carried out by hand exactly as all_to_all_single /
all_gather_into_tensor define it from the ranks' split tables; each rank then scatters and fuses (`finish`). The fused tokens
and fusion weights must be BIT-EQUAL to the plain single-GPU path on the same videos -- throughput form (every rank fuses
its own videos) with both exchanges, and the latency form (one video split over all ranks by LanguageBind clip and DINOv2 /
SigLIP frame ranges, every rank fuses). What stays unverified without a multi-GPU box is only RCCL's transport itself."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_path(dev):
    """merv-full geometry at 2 blocks per encoder (the placement logic does not depend on depth)."""
    from merv_amd.backbones import random_weights
    from merv_amd.encoder import merv_full_specs
    from merv_amd.projector import CrossAttentionAdapterLearnableQuery
    from merv_amd.visual_path import MervVisualPath
    specs = [dataclasses.replace(s, layers=2) for s in merv_full_specs()]
    Ws = [random_weights(s, 50 + i, device=dev, bf16_exact=True) for i, s in enumerate(specs)]
    g = torch.Generator(device=dev).manual_seed(7)
    pw = [((torch.randn(4096, s.dim, generator=g, device=dev) * s.dim**-0.5).to(torch.bfloat16).float(),
           torch.randn(4096, generator=g, device=dev) * 0.02) for s in specs]
    torch.manual_seed(1024)
    fusion = CrossAttentionAdapterLearnableQuery(embed_dim=3072, llm_dim=4096, token_length=1024, averagetoken=True)
    return specs, MervVisualPath(specs, Ws, pw, fusion, dev)


def _whole_videos(specs, dev, seed, v0, n):
    """The pixels DistributedVisualPath.synth_unit_pixels slices its units from."""
    pix = []
    for e, sp in enumerate(specs):
        vids = [torch.randn(sp.pixel_shape(1), generator=torch.Generator(device=dev).manual_seed(seed * 1000003 + e * 10007 + v),
                            device=dev).to(torch.bfloat16) for v in range(v0, v0 + n)]
        pix.append(torch.cat(vids, 0))
    return pix


def _run_world(path, specs, world, exchange, seed, **kw):
    from merv_amd.distributed import DistributedVisualPath
    ranks = [DistributedVisualPath(path, specs, world, r, exchange=exchange, **kw) for r in range(world)]
    sends = []
    for dp in ranks:  # phase 1 on every rank
        send = dp.produce(dp.synth_unit_pixels(seed))
        torch.cuda.synchronize()
        sends.append(send.clone())
    outs = []
    for r, dp in enumerate(ranks):  # the collective, then phase 2
        if exchange == "all_to_all":
            # all_to_all_single: rank r receives, from every source p in rank order, the slice of p's send buffer addressed to r
            parts = []
            for p, src in enumerate(ranks):
                off = sum(src.send_splits[:r])
                parts.append(sends[p][off:off + src.send_splits[r]])
                assert src.send_splits[r] == dp.recv_splits[p]
            recv = torch.cat(parts, 0)
        else:  # all_gather_into_tensor: every rank's (padded) buffer, in rank order
            assert all(s.shape[0] == dp.max_rows for s in sends)
            recv = torch.cat(sends, 0)
        fused, w = dp.finish(recv)
        torch.cuda.synchronize()
        outs.append((fused.clone(), w.clone()))
    return ranks, outs


@pytest.mark.parametrize("world,videos_per_rank", [(2, 1), (2, 2), (4, 1), (8, 1)])
@pytest.mark.parametrize("exchange", ["all_to_all", "all_gather"])
def test_throughput_form_equals_single_gpu_path(dev, small_path, world, videos_per_rank, exchange):
    specs, path = small_path
    ranks, outs = _run_world(path, specs, world, exchange, seed=11, videos_per_rank=videos_per_rank)
    for r, (fused, w) in enumerate(outs):
        ref_f, ref_w = path.forward(_whole_videos(specs, dev, 11, r * videos_per_rank, videos_per_rank))
        torch.cuda.synchronize()
        assert torch.equal(fused, ref_f), (world, exchange, r)
        assert torch.equal(w, ref_w)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_latency_form_one_video_over_all_ranks(dev, small_path, world):
    specs, path = small_path
    ranks, outs = _run_world(path, specs, world, "all_gather", seed=12, n_videos=1, replicate_fusion=True)
    # the plan really splits inside the video: LanguageBind by clip and / or DINOv2 / SigLIP by frame ranges; ViViT never
    units = [u for dp in ranks for u in dp.my_units]
    assert any(u[0] != 2 and (u[3], u[4]) != (0, specs[u[0]].frames) for u in units)
    assert all((u[3], u[4]) == (0, specs[2].frames) for u in units if u[0] == 2)
    ref_f, ref_w = path.forward(_whole_videos(specs, dev, 12, 0, 1))
    torch.cuda.synchronize()
    for fused, w in outs:  # every rank holds the fused tokens of the video
        assert torch.equal(fused, ref_f) and torch.equal(w, ref_w)


def test_literal_one_encoder_per_gpu_placement(dev, small_path):
    specs, path = small_path
    ranks, outs = _run_world(path, specs, 4, "all_gather", seed=13, n_videos=1, replicate_fusion=True, placement="per_encoder")
    assert [dp.my_units[0][0] for dp in ranks] == [0, 1, 2, 3]
    ref_f, ref_w = path.forward(_whole_videos(specs, dev, 13, 0, 1))
    torch.cuda.synchronize()
    for fused, w in outs:
        assert torch.equal(fused, ref_f) and torch.equal(w, ref_w)
