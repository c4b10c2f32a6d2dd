"""CPU, world_size 2, gloo: the multi-GPU placement + exchange logic of DistributedVisualPath with a stand-in compute
object (the product compute is HIP-only). Checks that every rank ends up with exactly the fused result a single
process computes for its own videos, for both exchange modes, for the one-video latency placement that splits
LanguageBind by clip and DINOv2 / SigLIP by frame range inside a video, and for the one-encoder-per-rank placement."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
T, C = 16, 8  # projected rows per video, width


class FakeSpec:
    """The fields DistributedVisualPath / plan_units read from merv_amd.encoder.EncoderSpec."""

    def __init__(self, name, cost, frames, tubelet=1, temporal_frames=0, joint_space_time=False, pix_layout="BFCHW"):
        self.name, self._cost, self.frames, self.tubelet = name, cost, frames, tubelet
        self.temporal_frames, self.joint_space_time, self.pix_layout = temporal_frames, joint_space_time, pix_layout

    def flops_per_video(self):
        return self._cost

    def pixel_shape(self, batch):
        return (batch, 3, self.frames) if self.pix_layout == "BCFHW" else (batch, self.frames, 3)


# merv-full's structure in miniature: LanguageBind in clips of 8 (channel-first pixels), DINOv2 / SigLIP per frame,
# ViViT whole (tubelets of 2: 32 frames -> 16 output frames)
SPECS = [FakeSpec("languagebind", 3.281, 16, temporal_frames=8, pix_layout="BCFHW"), FakeSpec("dinov2", 2.525, 16),
         FakeSpec("vivit", 0.903, 32, tubelet=2, joint_space_time=True), FakeSpec("siglip", 0.513, 16)]


class FakeLocal:
    """Frame-separable stand-in compute: the projected rows of an output frame depend on that frame's pixels only (as in
    the product: per-frame sequences, pooling inside a frame); fuse = plain mean over encoders."""
    device = torch.device("cpu")
    dtype = torch.float32
    T_vis, llm_dim = T, C

    def encode_project(self, e, pix, stream=None, out=None, frames=None):
        s = SPECS[e]
        x = pix.float()
        x = x.permute(0, 2, 1) if s.pix_layout == "BCFHW" else x  # -> [n, F, 3]
        n, F = x.shape[0], x.shape[1]
        assert F == (frames if frames is not None else s.frames), (F, frames, s.frames)
        fo = F // s.tubelet
        per = T // (s.frames // s.tubelet)  # rows per output frame
        val = x.reshape(n, fo, s.tubelet * 3).sum(-1) * (e + 1)  # [n, fo]
        ramp = torch.arange(per * C, dtype=torch.float32).reshape(1, 1, per, C) * 0.01 * (e + 1)
        return (val[:, :, None, None] + ramp).reshape(n, fo * per, C)

    def fuse(self, V):
        st = torch.stack(V, 0)
        return st.mean(0), torch.full((st.shape[1], st.shape[0]), 1.0 / st.shape[0])


def _video_pixels(e, v):
    s = SPECS[e]
    g = torch.Generator().manual_seed(1000 * e + v)
    return torch.randn(s.pixel_shape(1), generator=g)


def _unit_pixels(unit):
    e, v0, v1, f0, f1 = unit
    s = SPECS[e]
    pix = torch.cat([_video_pixels(e, v) for v in range(v0, v1)], 0)
    if (f0, f1) != (0, s.frames):
        pix = (pix[:, :, f0:f1] if s.pix_layout == "BCFHW" else pix[:, f0:f1]).contiguous()
    return pix


def _worker(rank, world, port, kw, q):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from merv_amd.distributed import DistributedVisualPath
        local = FakeLocal()
        d = DistributedVisualPath(local, SPECS, world, rank, **kw)
        fused, w = d.forward([_unit_pixels(u) for u in d.my_units])
        # single-process expectation for the videos this rank fuses: whole, unsplit forwards
        mine = range(d.G) if d.replicate else range(rank * d.B, (rank + 1) * d.B)
        V = [torch.cat([local.encode_project(e, _video_pixels(e, v)) for v in mine], 0) for e in range(len(SPECS))]
        ref, _ = local.fuse(V)
        q.put((rank, bool(torch.equal(fused, ref)), [list(u) for u in d.my_units], d.exchange_bytes_per_rank()))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kw, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _covered(res):
    return sorted((e, v, f) for (_, _, us, _) in res for (e, v0, v1, f0, f1) in us for v in range(v0, v1) for f in range(f0, f1))


@pytest.mark.parametrize("exchange", ["all_to_all", "all_gather"])
@pytest.mark.parametrize("per_rank", [1, 3])
def test_two_rank_exchange_matches_single_process(exchange, per_rank):
    """Throughput form: every rank ends with exactly the fused result one process computes for ITS videos."""
    world = 2
    res = _run(world, dict(videos_per_rank=per_rank, exchange=exchange))
    assert all(ok for (_, ok, _, _) in res), res
    assert _covered(res) == sorted((e, v, f) for e, s in enumerate(SPECS) for v in range(world * per_rank) for f in range(s.frames))


def test_two_rank_split_inside_one_video_matches_unsplit():
    """Latency form: ONE video over two ranks -- an encoder is split INSIDE the video (DINOv2 by frame range
    here; LanguageBind only at clip boundaries, ViViT never), rows all-gathered, every rank fuses: both ranks hold the
    unsplit single-process result bit for bit."""
    res = _run(2, dict(n_videos=1, replicate_fusion=True, exchange="all_gather"))
    assert all(ok for (_, ok, _, _) in res), res
    units = [tuple(u) for (_, _, us, _) in res for u in us]
    partial = [u for u in units if (u[3], u[4]) != (0, SPECS[u[0]].frames)]
    assert partial, units  # the plan really split an encoder inside the video
    for (e, v0, v1, f0, f1) in partial:
        assert not SPECS[e].joint_space_time and f0 % max(SPECS[e].temporal_frames, 1) == 0 and f1 % max(SPECS[e].temporal_frames, 1) == 0
    assert _covered(res) == sorted((e, 0, f) for e, s in enumerate(SPECS) for f in range(s.frames))


def test_two_rank_one_encoder_per_rank_placement():
    """The literal configs[2] placement (encoder e on rank e % world) through the same exchange."""
    res = _run(2, dict(videos_per_rank=2, exchange="all_gather", placement="per_encoder"))
    assert all(ok for (_, ok, _, _) in res), res
    for (rank, _, us, _) in res:
        assert {u[0] % 2 for u in us} == {rank}


def test_plan_units_covers_and_balances():
    from merv_amd.visual_path import plan_units
    costs = [s.flops_per_video() for s in SPECS]
    frames = [s.frames for s in SPECS]
    atoms = [8, 1, 32, 1]
    for world, n in [(4, 1), (8, 1), (2, 1), (2, 16), (8, 64), (3, 5), (4, 4)]:
        plan = plan_units(costs, n, world, frames, atoms)
        seen = set()
        for r in plan:
            per_enc = {}
            for (e, v0, v1, f0, f1) in r:
                assert f0 % atoms[e] == 0 and f1 % atoms[e] == 0 and (v1 - v0 == 1 or (f0, f1) == (0, frames[e]))
                per_enc.setdefault(e, []).append((v0, f0))
                for v in range(v0, v1):
                    for f in range(f0, f1):
                        assert (e, v, f) not in seen
                        seen.add((e, v, f))
        assert len(seen) == n * sum(frames)
        loads = [sum(costs[e] * (v1 - v0) * (f1 - f0) / frames[e] for (e, v0, v1, f0, f1) in r) for r in plan]
        if n >= world * 8:
            assert max(loads) / (sum(loads) / world) < 1.02
    one = plan_units(costs, 1, 4, frames, atoms)
    assert max(sum(costs[e] * (f1 - f0) / frames[e] for (e, _, _, f0, f1) in r) for r in one) < 1.95  # vs 3.28 one-encoder-per-GPU


# ---- row f-4: data-parallel gradient exchange (FlatGradSync) ----
def _grad_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from merv_amd.train import FlatGradSync
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        frozen = torch.nn.Linear(3, 3).requires_grad_(False)
        sync = FlatGradSync(list(model.parameters()) + list(frozen.parameters()), bucket_bytes=64)  # 16 floats: several buckets
        x = torch.arange(24, dtype=torch.float32).reshape(4, 6) / 10
        y = torch.arange(12, dtype=torch.float32).reshape(4, 3) / 7
        lo, hi = rank * 2, rank * 2 + 2
        for _ in range(2):  # two micro-batches accumulate into the same flat buffer
            (torch.nn.functional.mse_loss(frozen(model(x[lo:hi])), y[lo:hi]) / 2).backward()
        sync.all_reduce_mean()
        norm = float(sync.clip_grad_norm_(1e9))
        # single-process expectation: mean over ranks of each rank's accumulated gradient
        ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        ref.load_state_dict(model.state_dict())
        for r in range(world):
            (torch.nn.functional.mse_loss(frozen(ref(x[r * 2:r * 2 + 2])), y[r * 2:r * 2 + 2]) / world).backward()
        ok = all(torch.allclose(p.grad, pr.grad, atol=1e-6) for p, pr in zip(model.parameters(), ref.parameters()))
        ref_norm = float(torch.cat([p.grad.flatten() for p in ref.parameters()]).norm())
        sync.check_views()
        n_param = sum(p.numel() for p in model.parameters())
        q.put((rank, ok, abs(norm - ref_norm) < 1e-5, sync.flat.numel() == n_param))
    finally:
        dist.destroy_process_group()


def test_two_rank_flat_grad_sync_matches_single_process():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] and r[2] and r[3] for r in res), res
