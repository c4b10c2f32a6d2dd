"""
DistributedVisualPath: the visual path over N MI355X, one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no multi-GPU inference path for this branch (the only collectives are FSDP's);
this is the branch-parallel placement BASELINE.json's north_star asks for, generalised so that it stays balanced:

  * work unit = (encoder e, video v) of the step's global batch of G = world * videos_per_rank videos; unit cost =
    the encoder's FLOPs. `plan_units` (visual_path.py) gives every rank, per encoder, one contiguous run of videos,
    so a rank runs at most one batched forward per encoder. With world=4, G=1 this degenerates to the north_star's
    "one encoder per GPU"; larger batches even out the 6.4x LanguageBind/SigLIP cost ratio.
  * every rank holds all encoder weights (1.75 GB bf16 of 288 GB HBM): placement never moves weights.
  * exchange: each projected unit [1024, llm] bf16 (8.39 MB) must reach the rank that fuses video v (owner(v) =
    v // videos_per_rank, the rank whose LLM replica prefills it).
      - "all_to_all": one all_to_all_single; every unit travels exactly once over the direct xGMI link
        between producer and owner (7 point-to-point links per GPU, no ring hop).
      - "all_gather": the literal north_star collective; every rank receives everything (world x the bytes).
    then every rank runs the fusion kernels for its own videos.

The compute object is injected (`local`: encode_project(e, pixels) and fuse(list)), so the placement / exchange
logic is testable on CPU with the gloo backend and a stand-in compute (tests/test_distributed_cpu.py).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist

from .visual_path import plan_units


class DistributedVisualPath:
    def __init__(self, local, costs: Sequence[float], world: int, rank: int, videos_per_rank: int,
                 exchange: str = "all_to_all", group=None):
        if exchange not in ("all_to_all", "all_gather"):
            raise ValueError(f"unknown exchange `{exchange}`")
        self.local = local
        self.world, self.rank, self.B = world, rank, videos_per_rank
        self.G = world * videos_per_rank
        self.E = len(costs)
        self.exchange = exchange
        self.group = group
        self.plan = plan_units(costs, self.G, world)
        self.my_units = self.plan[rank]
        # flat unit order per producer rank: the order in which it lays units out in its send buffer,
        # sorted by owner rank so that all_to_all_single can use contiguous splits
        self.send_order: List[List[Tuple[int, int]]] = []  # per rank: [(e, v)] sorted by (owner, e, v)
        for r in range(world):
            units = [(e, v) for (e, v0, v1) in self.plan[r] for v in range(v0, v1)]
            units.sort(key=lambda ev: (ev[1] // videos_per_rank, ev[0], ev[1]))
            self.send_order.append(units)
        self.send_splits = [sum(1 for (_, v) in self.send_order[rank] if v // videos_per_rank == o) for o in range(world)]
        self.recv_splits = [sum(1 for (_, v) in self.send_order[p] if v // videos_per_rank == rank) for p in range(world)]
        # where each of my videos' (e, v) lands in my receive buffer
        self.recv_index: Dict[Tuple[int, int], int] = {}
        pos = 0
        for p in range(world):
            for (e, v) in self.send_order[p]:
                if v // videos_per_rank == rank:
                    self.recv_index[(e, v)] = pos
                    pos += 1
        assert pos == self.E * self.B, "every (encoder, video) of my videos must arrive exactly once"
        self.max_units = max(len(u) for u in self.send_order)

    # ---- synthetic inputs: each rank only materialises the pixels of its own units -------------------------------
    def synth_unit_pixels(self, specs, seed: int) -> List[torch.Tensor]:
        dev = self.local.device
        out = []
        for (e, v0, v1) in self.my_units:
            g = torch.Generator(device=dev).manual_seed(seed * 1000003 + e * 10007 + v0)
            out.append(torch.randn(specs[e].pixel_shape(v1 - v0), generator=g, device=dev).to(torch.bfloat16))
        return out

    def forward(self, unit_pixels: Sequence[torch.Tensor]):
        """unit_pixels[i] = pixels of self.my_units[i] (encoder e's layout, videos v0..v1).
        Returns (fused [B, T, llm], weights [B, E]) for this rank's own videos."""
        local = self.local
        produced: Dict[Tuple[int, int], torch.Tensor] = {}
        streams = getattr(local, "streams", None)
        main = torch.cuda.current_stream(local.device) if streams else None
        if streams:
            start = torch.cuda.Event()
            start.record(main)
        for i, ((e, v0, v1), pix) in enumerate(zip(self.my_units, unit_pixels)):
            if streams:
                st = streams[i % len(streams)]
                st.wait_event(start)
                proj = local.encode_project(e, pix, st)
                ev = torch.cuda.Event()
                ev.record(st)
                main.wait_event(ev)
            else:
                proj = local.encode_project(e, pix)
            for j, v in enumerate(range(v0, v1)):
                produced[(e, v)] = proj[j]
        T, Cc = local.T_vis, local.llm_dim
        dt, dev = getattr(local, "dtype", torch.bfloat16), local.device
        if self.exchange == "all_to_all":
            send = torch.empty(len(self.send_order[self.rank]), T, Cc, dtype=dt, device=dev)
            for i, ev in enumerate(self.send_order[self.rank]):
                send[i].copy_(produced[ev])
            recv = torch.empty(self.E * self.B, T, Cc, dtype=dt, device=dev)
            dist.all_to_all_single(recv, send, output_split_sizes=self.recv_splits, input_split_sizes=self.send_splits,
                                   group=self.group)
            def fetch(e, v):
                return recv[self.recv_index[(e, v)]]
        else:
            send = torch.zeros(self.max_units, T, Cc, dtype=dt, device=dev)
            for i, ev in enumerate(self.send_order[self.rank]):
                send[i].copy_(produced[ev])
            gathered = torch.empty(self.world * self.max_units, T, Cc, dtype=dt, device=dev)
            dist.all_gather_into_tensor(gathered, send, group=self.group)
            where = {ev: p * self.max_units + i for p in range(self.world) for i, ev in enumerate(self.send_order[p])}
            def fetch(e, v):
                return gathered[where[(e, v)]]
        mine = range(self.rank * self.B, (self.rank + 1) * self.B)
        V = [torch.stack([fetch(e, v) for v in mine], 0) for e in range(self.E)]
        return local.fuse(V)
