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
