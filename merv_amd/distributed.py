"""
DistributedVisualPath: the visual path over N MI355X, one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no multi-GPU inference path for this branch (the only collectives are FSDP's);
this is the branch-parallel placement BASELINE.json's north_star asks for, generalised so that it stays balanced.
NOT YET RUN ON MORE THAN ONE RANK OF REAL HARDWARE (no multi-GPU box was available to the builder): covered by
world-size-2 gloo tests with a stand-in compute (tests/test_distributed_cpu.py) and by world-size-1 RCCL runs.

  * work unit = (encoder e, videos [v0, v1), frames [f0, f1)) -- `plan_units` (visual_path.py). The independent pieces of
    the path: whole videos for ViViT (joint space-time attention), clips of 8 frames for LanguageBind
    (modeling_video.py:140-146), single frames for DINOv2 / SigLIP (dinov2_video.py:135-136). The 3davg projector pools
    inside a frame only (nn_utils.py:320-330 with output_frames == T), so a frame range of an encoder yields exactly the
    matching row range [f0/tt*64, f1/tt*64) of that video's projected [1024, llm] tokens.
      - throughput form: G = world * videos_per_rank videos per step, video v is fused on rank v // videos_per_rank (the
        rank whose LLM replica prefills it);
      - latency form (`n_videos=1, replicate_fusion=True`): ONE video spread over all ranks, every rank receives all rows
        and fuses, so each holds the fused tokens for its prefill (4 ranks: makespan 1.89 TFLOP vs 3.28 for one encoder per
        GPU; 8 ranks: 1.64, the LanguageBind clip);
      - `placement="per_encoder"`: the literal configs[2] form, encoder e on rank e % world.
  * every rank holds all encoder weights (1.75 GB bf16 of 288 GB HBM): placement never moves weights.
  * exchange of projected rows (bf16, llm wide):
      - "all_to_all": one all_to_all_single with uneven row splits; every row travels exactly once over the direct xGMI
        link between producer and owner (7 point-to-point links per GPU, no ring hop);
      - "all_gather": the literal north_star collective (all_gather_into_tensor of each rank's rows padded to the largest
        rank's count); every rank receives everything (world x the bytes).
    then every owner runs the fusion kernels for its videos.

The compute object is injected (`local`: encode_project(e, pixels, stream, frames=) and fuse(list)), so the placement /
exchange logic is testable on CPU with the gloo backend and a stand-in compute.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .visual_path import plan_one_encoder_per_rank, plan_units

Unit = Tuple[int, int, int, int, int]  # (e, v0, v1, f0, f1)


def atom_frames(spec) -> int:
    """Smallest independent frame count of an encoder (see plan_units)."""
    if spec.joint_space_time:
        return spec.frames
    return spec.temporal_frames if spec.temporal_frames else spec.tubelet


class DistributedVisualPath:
    def __init__(self, local, specs: Sequence, world: int, rank: int, videos_per_rank: Optional[int] = None,
                 exchange: str = "all_to_all", group=None, n_videos: Optional[int] = None, replicate_fusion: bool = False,
                 placement: str = "balanced"):
        if exchange not in ("all_to_all", "all_gather"):
            raise ValueError(f"unknown exchange `{exchange}`")
        if placement not in ("balanced", "per_encoder"):
            raise ValueError(f"unknown placement `{placement}`")
        if replicate_fusion and exchange != "all_gather":
            raise ValueError("replicate_fusion (every rank fuses every video) needs the all_gather exchange")
        # two forms, nothing in between: throughput (videos_per_rank, every rank fuses its own videos) or latency (n_videos
        # spread over all ranks, every rank fuses all of them)
        if replicate_fusion:
            if n_videos is None or videos_per_rank is not None:
                raise ValueError("latency form: give n_videos (and no videos_per_rank) together with replicate_fusion")
        elif videos_per_rank is None or n_videos is not None:
            raise ValueError("throughput form: give videos_per_rank (and no n_videos); n_videos needs replicate_fusion")
        self.local, self.specs = local, list(specs)
        self.world, self.rank = world, rank
        self.E = len(self.specs)
        self.exchange, self.group, self.replicate = exchange, group, replicate_fusion
        self.frames = [s.frames for s in self.specs]
        self.G = n_videos if n_videos is not None else world * videos_per_rank
        self.B = self.G if replicate_fusion else videos_per_rank  # videos this rank fuses
        costs = [float(s.flops_per_video()) for s in self.specs]
        if placement == "per_encoder":
            self.plan = plan_one_encoder_per_rank(self.E, self.G, world, self.frames)
        else:
            self.plan = plan_units(costs, self.G, world, self.frames, [atom_frames(s) for s in self.specs])
        self.my_units: List[Unit] = self.plan[rank]
        self.T, self.C = local.T_vis, local.llm_dim
        for e in range(self.E):
            if (self.T * atom_frames(self.specs[e])) % self.frames[e]:
                raise ValueError(f"encoder {e}: {self.T} projected rows do not split over atoms of {atom_frames(self.specs[e])} frames")

        # chunk = the rows one unit contributes to ONE video: (e, v, f0, f1); per producer rank, in send order
        def chunks_of(r: int):
            ch = [(e, v, f0, f1) for (e, v0, v1, f0, f1) in self.plan[r] for v in range(v0, v1)]
            ch.sort(key=lambda c: (self.owner(c[1]), c[0], c[1], c[2]))
            return ch

        self.send_order = [chunks_of(r) for r in range(world)]
        self.rows_sent = [sum(self.rows(c) for c in self.send_order[r]) for r in range(world)]
        self.max_rows = max(self.rows_sent)
        if not replicate_fusion:
            self.send_splits = [sum(self.rows(c) for c in self.send_order[rank] if self.owner(c[1]) == o) for o in range(world)]
            self.recv_splits = [sum(self.rows(c) for c in self.send_order[p] if self.owner(c[1]) == rank) for p in range(world)]
        # where every chunk this rank fuses lands: all_to_all -> offset in the receive buffer; all_gather -> p * max_rows + offset
        self.recv_at: List[Tuple[Tuple[int, int, int, int], int]] = []
        pos_a2a = 0
        for p in range(world):
            off = 0
            for c in self.send_order[p]:
                if self.mine(c[1]):
                    self.recv_at.append((c, pos_a2a if exchange == "all_to_all" else p * self.max_rows + off))
                    pos_a2a += self.rows(c)
                off += self.rows(c)
        got = sum(self.rows(c) for c, _ in self.recv_at)
        assert got == self.E * self.B * self.T, "every projected row of my videos must arrive exactly once"
        self._bufs: Dict[str, torch.Tensor] = {}

    # ---- geometry -----------------------------------------------------------------------------------------------
    def owner(self, v: int) -> int:
        return 0 if self.replicate else v // self.B

    def mine(self, v: int) -> bool:
        return True if self.replicate else self.owner(v) == self.rank

    def rows(self, c) -> int:
        e, _, f0, f1 = c
        return self.T * (f1 - f0) // self.frames[e]

    def row0(self, c) -> int:
        e, _, f0, _ = c
        return self.T * f0 // self.frames[e]

    def exchange_bytes_per_rank(self) -> Dict[str, int]:
        es = 2 if getattr(self.local, "dtype", torch.bfloat16) == torch.bfloat16 else 4
        if self.exchange == "all_to_all":
            sent = sum(s for o, s in enumerate(self.send_splits) if o != self.rank)
            recv = sum(s for p, s in enumerate(self.recv_splits) if p != self.rank)
        else:
            sent = self.max_rows * (self.world - 1)
            recv = self.max_rows * (self.world - 1)
        return {"sent": sent * self.C * es, "received": recv * self.C * es}

    def describe_plan(self) -> List[str]:
        names = [getattr(s, "name", str(i)) for i, s in enumerate(self.specs)]
        return ["rank %d: " % r + (", ".join(
            f"{names[e]} v{v0}" + (f"-{v1 - 1}" if v1 - v0 > 1 else "") + ("" if (f0, f1) == (0, self.frames[e]) else f" f{f0}-{f1 - 1}")
            for (e, v0, v1, f0, f1) in units) or "idle") for r, units in enumerate(self.plan)]

    # ---- synthetic inputs: each rank only materialises the pixels of its own units; a video's pixels depend on
    # (seed, e, v) alone, so every placement of the same global batch sees the same data ------------------------------
    def synth_unit_pixels(self, seed: int, dtype=torch.bfloat16) -> List[torch.Tensor]:
        dev = self.local.device
        out = []
        for (e, v0, v1, f0, f1) in self.my_units:
            s = self.specs[e]
            vids = []
            for v in range(v0, v1):
                g = torch.Generator(device=dev).manual_seed(seed * 1000003 + e * 10007 + v)
                vids.append(torch.randn(s.pixel_shape(1), generator=g, device=dev).to(dtype))
            pix = torch.cat(vids, 0)
            if (f0, f1) != (0, s.frames):
                pix = (pix[:, :, f0:f1] if s.pix_layout == "BCFHW" else pix[:, f0:f1]).contiguous()
            out.append(pix)
        return out

    def _buf(self, name: str, rows: int) -> torch.Tensor:
        t = self._bufs.get(name)
        if t is None or t.shape[0] != max(rows, 1):
            dt = getattr(self.local, "dtype", torch.bfloat16)
            t = self._bufs[name] = torch.zeros(max(rows, 1), self.C, dtype=dt, device=self.local.device)
        return t

    def forward(self, unit_pixels: Sequence[torch.Tensor]):
        """unit_pixels[i] = pixels of self.my_units[i] (encoder e's layout, videos v0..v1, frames f0..f1 only).
        Returns (fused [B, T, llm], weights [B, E]) for the videos this rank fuses."""
        send = self.produce(unit_pixels)
        # ---- one collective
        if self.exchange == "all_to_all":
            recv = self._buf("recv", self.E * self.B * self.T)
            dist.all_to_all_single(recv[: self.E * self.B * self.T], send[: self.rows_sent[self.rank]],
                                   output_split_sizes=self.recv_splits, input_split_sizes=self.send_splits, group=self.group)
        else:
            recv = self._buf("gathered", self.world * self.max_rows)
            dist.all_gather_into_tensor(recv, send, group=self.group)
        return self.finish(recv)

    def produce(self, unit_pixels: Sequence[torch.Tensor]) -> torch.Tensor:
        """This rank's units through encoder + projector, their projected rows packed in send order (the collective's input:
        rows_sent[rank] rows for all_to_all, padded to max_rows for all_gather)."""
        local = self.local
        if len(unit_pixels) != len(self.my_units):
            raise ValueError(f"expected {len(self.my_units)} unit pixel tensors, got {len(unit_pixels)}")
        produced: Dict[Tuple[int, int, int, int], torch.Tensor] = {}
        streams = getattr(local, "streams", None)
        main = torch.cuda.current_stream(local.device) if streams else None
        if streams:
            start = torch.cuda.Event()
            start.record(main)
            used = {}
        for (e, v0, v1, f0, f1), pix in zip(self.my_units, unit_pixels):
            nf = None if (f0, f1) == (0, self.frames[e]) else f1 - f0
            if streams:
                st = streams[e % len(streams)]  # units of one encoder share its workspace: same stream, in order
                if e not in used:
                    st.wait_event(start)
                proj = local.encode_project(e, pix, st, frames=nf)
                if len([u for u in self.my_units if u[0] == e]) > 1:
                    with torch.cuda.stream(st):
                        proj = proj.clone()  # the persistent per-shape buffer may be reused by this encoder's next unit
                used[e] = st
            else:
                proj = local.encode_project(e, pix, frames=nf)
            for j, v in enumerate(range(v0, v1)):
                produced[(e, v, f0, f1)] = proj[j]
        if streams:
            for st in used.values():
                ev = torch.cuda.Event()
                ev.record(st)
                main.wait_event(ev)
        T, C = self.T, self.C
        # ---- pack this rank's rows in send order
        send = self._buf("send", self.rows_sent[self.rank] if self.exchange == "all_to_all" else self.max_rows)
        off = 0
        for c in self.send_order[self.rank]:
            n = self.rows(c)
            send[off:off + n].copy_(produced[c])
            off += n
        return send

    def finish(self, recv: torch.Tensor):
        """The collective's output -> the chunks of my videos scattered into V_e [B, T, llm] -> fusion."""
        local, T, C = self.local, self.T, self.C
        V = [self._buf(f"V{e}", self.B * T).view(self.B, T, C) for e in range(self.E)]
        v_base = 0 if self.replicate else self.rank * self.B
        for c, at in self.recv_at:
            e, v, _, _ = c
            r0, n = self.row0(c), self.rows(c)
            V[e][v - v_base, r0:r0 + n].copy_(recv[at:at + n])
        return local.fuse(V)
