"""CPU, world_size 2, gloo: the multi-GPU placement + exchange logic of DistributedVisualPath with a stand-in compute
object (the product compute is HIP-only). Checks that every rank ends up with exactly the fused result a single
process computes for its own videos, for both exchange modes."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
COSTS = [3.281, 2.525, 0.903, 0.513]
T, C = 4, 8


class FakeLocal:
    """encode_project(e, pix) = deterministic function of (e, pixel values); fuse = plain mean over encoders."""
    device = torch.device("cpu")
    dtype = torch.float32
    T_vis, llm_dim = T, C

    def encode_project(self, e, pix, stream=None):
        # pix: [n, 3] "pixels" -> [n, T, C]
        base = pix.sum(-1)[:, None, None] * (e + 1)
        return base + torch.arange(T * C, dtype=torch.float32).reshape(1, T, C) * 0.01 * (e + 1)

    def fuse(self, V):
        st = torch.stack(V, 0)
        return st.mean(0), torch.full((st.shape[1], st.shape[0]), 1.0 / st.shape[0])


def _pixels(e, v0, v1):
    return torch.stack([torch.tensor([v + 1.0, e * 0.5, (v * 7 + e) % 3 * 1.0]) for v in range(v0, v1)], 0)


def _worker(rank, world, port, per_rank, exchange, q):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from merv_amd.distributed import DistributedVisualPath
        local = FakeLocal()
        d = DistributedVisualPath(local, COSTS, world, rank, per_rank, exchange=exchange)
        unit_pixels = [_pixels(e, v0, v1) for (e, v0, v1) in d.my_units]
        fused, w = d.forward(unit_pixels)
        # single-process expectation for this rank's videos
        mine = range(rank * per_rank, (rank + 1) * per_rank)
        V = [torch.cat([local.encode_project(e, _pixels(e, v, v + 1)) for v in mine], 0) for e in range(len(COSTS))]
        ref, _ = local.fuse(V)
        q.put((rank, bool(torch.equal(fused, ref)), [list(u) for u in d.my_units]))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("exchange", ["all_to_all", "all_gather"])
@pytest.mark.parametrize("per_rank", [1, 3])
def test_two_rank_exchange_matches_single_process(exchange, per_rank):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, exchange, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for (_, ok, _) in res), res
    units = sorted(tuple(u) for (_, _, us) in res for u in us)
    covered = {(e, v) for (e, v0, v1) in units for v in range(v0, v1)}
    assert covered == {(e, v) for e in range(4) for v in range(world * per_rank)}


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
