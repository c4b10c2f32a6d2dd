"""
MervVisualPath: the visual branch of MERV.forward (merv/models/vidlms/merv.py:562-609) on MI355X --
E encoders -> reshape [B,T,S,C] -> 3davg+linear projectors -> cross-encoder fusion -> [B, 1024, llm] fused tokens.

Single GPU: the encoders are independent until fusion (merv.py:563-566 is a list comprehension with no cross-talk),
so each runs on its own HIP stream, event-joined before the fusion kernels (the reference runs them one after the
other on one stream).

Multi GPU (torch.distributed over RCCL/xGMI, one process per GPU): the (encoder, video) work units of a global batch
are placed on ranks by an LPT schedule over their FLOP cost (every rank holds all encoder weights: 1.75 GB bf16 of
288 GB); each rank projects its units and ONE all-gather of the projected [*, 1024, llm] bf16 tokens hands every
rank the full V_e tensors, after which every rank fuses the videos it owns. See `plan_units`.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import check, ptr
from .encoder import EncoderSpec, HipEncoder
from .projector import CrossAttentionAdapterLearnableQuery


class MervVisualPath:
    def __init__(self, specs: Sequence[EncoderSpec], enc_weights: Sequence[Dict],
                 proj_weights: Sequence[Tuple[torch.Tensor, torch.Tensor]], fusion: CrossAttentionAdapterLearnableQuery,
                 device, out_size: int = 8, concurrent_streams: bool = True):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MervVisualPath needs a ROCm device; merv_amd has no CPU path")
        self.lib = _lib.load()
        self.specs = list(specs)
        self.encoders = [HipEncoder(s, w, self.device) for s, w in zip(specs, enc_weights)]
        self.proj = [(w.detach().to(self.device, torch.bfloat16).contiguous(),
                      b.detach().to(self.device, torch.float32).contiguous()) for w, b in proj_weights]
        self.llm_dim = self.proj[0][0].shape[0]
        self.out_size = out_size
        self.fusion = fusion
        self.fusion.prepare(self.device)
        self.tokens_out = {s.t_out * out_size * out_size for s in specs}
        if len(self.tokens_out) != 1:  # merv.py:175-193 consistency assert
            raise ValueError(f"Output token length is not consistent across projectors: {self.tokens_out}")
        self.T_vis = self.tokens_out.pop()
        self.concurrent = concurrent_streams and len(self.encoders) > 1
        self.streams = [torch.cuda.Stream(self.device) for _ in self.encoders]
        self._bufs: Dict[Tuple[int, int], Dict[str, torch.Tensor]] = {}

    # -- buffers are persistent per (encoder, batch): no allocator traffic and no cross-stream lifetime issues
    def _enc_bufs(self, i: int, B: int) -> Dict[str, torch.Tensor]:
        key = (i, B)
        if key not in self._bufs:
            s = self.specs[i]
            o = self.out_size
            self._bufs[key] = {
                "tokens": torch.empty(B, s.num_patches, s.dim, dtype=torch.bfloat16, device=self.device),
                "pooled": torch.empty(B * s.t_out * o * o, s.dim, dtype=torch.bfloat16, device=self.device),
                "proj": torch.empty(B, self.T_vis, self.llm_dim, dtype=torch.bfloat16, device=self.device),
            }
            self.encoders[i].workspace(B)
        return self._bufs[key]

    def encode_project(self, i: int, pixels: torch.Tensor, stream: torch.cuda.Stream,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """a4-a9 for encoder i on `stream`: pixels -> projected [B, T_vis, llm] bf16."""
        s = self.specs[i]
        B = pixels.shape[0]
        bufs = self._enc_bufs(i, B)
        tok = self.encoders[i].forward(pixels, out=bufs["tokens"], stream=stream)
        dst = out if out is not None else bufs["proj"]
        w, b = self.proj[i]
        rc = self.lib.merv_projector_forward(ptr(tok), B, s.t_out, s.hp, s.dim, self.out_size, ptr(w), ptr(b),
                                             self.llm_dim, ptr(bufs["pooled"]), ptr(dst), stream.cuda_stream)
        check(rc, "merv_projector_forward")
        return dst

    def fuse(self, projected: Sequence[torch.Tensor]):
        return self.fusion(list(projected))

    def forward(self, pixels: Sequence[torch.Tensor]):
        """pixels[i]: encoder i's post-transform tensor. Returns (fused [B,T_vis,llm] bf16, weights [B,E] fp32)."""
        if len(pixels) != len(self.encoders):
            raise ValueError(f"expected {len(self.encoders)} pixel tensors, got {len(pixels)}")
        main = torch.cuda.current_stream(self.device)
        projected = []
        if self.concurrent:
            start = torch.cuda.Event()
            start.record(main)
            for i, pix in enumerate(pixels):
                st = self.streams[i]
                st.wait_event(start)
                projected.append(self.encode_project(i, pix, st))
                done = torch.cuda.Event()
                done.record(st)
                main.wait_event(done)
        else:
            for i, pix in enumerate(pixels):
                projected.append(self.encode_project(i, pix, main))
        return self.fuse(projected)


    def capture(self, pixels: Sequence[torch.Tensor]):
        """Record one forward for these input shapes into a hipGraph (torch.cuda.CUDAGraph: the library's launches go
        to torch's capturing stream, the per-encoder side streams fork from and re-join it through events, so the four
        encoder chains stay concurrent inside the graph). Returns `replay(new_pixels=None) -> (fused, weights)`; the
        outputs are the same tensors on every replay. At batch 1 a step is ~700 launches of 10-40 us kernels, so the
        graph removes most of the launch overhead; at batch 8 it is within noise."""
        self.forward(pixels)  # warm-up outside the capture: kernel attributes, workspaces, persistent buffers
        torch.cuda.synchronize(self.device)
        static = [p.clone() for p in pixels]
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fused, weights = self.forward(static)

        def replay(new_pixels: Optional[Sequence[torch.Tensor]] = None):
            if new_pixels is not None:
                for dst, src in zip(static, new_pixels):
                    dst.copy_(src)
            graph.replay()
            return fused, weights

        replay.graph = graph  # keep it alive with the callable
        return replay


# ------------------------------------------------------------------------------------------------------------
# multi-GPU placement
# ------------------------------------------------------------------------------------------------------------
def plan_units(costs: Sequence[float], n_videos: int, world: int) -> List[List[Tuple[int, int, int]]]:
    """Place (encoder e, video range [v0, v1)) work units on `world` ranks.

    costs[e] = FLOPs of encoder e per video. Units are whole (encoder, video) pairs; each rank receives, per
    encoder, one contiguous run of videos so that it runs ONE batched forward per encoder. Greedy fill in
    descending encoder cost: walk the ranks, give each the number of videos that brings it closest to the
    per-rank target without splitting a video. Returns per rank a list of (e, v0, v1). Deterministic; every
    (e, v) appears exactly once.
    """
    E = len(costs)
    total = sum(costs) * n_videos
    target = total / world
    load = [0.0] * world
    plan: List[List[Tuple[int, int, int]]] = [[] for _ in range(world)]
    order = sorted(range(E), key=lambda e: -costs[e])
    for e in order:
        v = 0
        # ranks sorted by current load (least loaded first), stable
        ranks = sorted(range(world), key=lambda r: (load[r], r))
        for j, r in enumerate(ranks):
            if v >= n_videos:
                break
            remaining_ranks = len(ranks) - j
            room = max(target - load[r], 0.0)
            k = int(round(room / costs[e])) if costs[e] > 0 else n_videos - v
            if remaining_ranks == 1:
                k = n_videos - v
            k = max(0, min(k, n_videos - v))
            if k == 0:
                continue
            plan[r].append((e, v, v + k))
            load[r] += k * costs[e]
            v += k
        if v < n_videos:  # rounding left a tail: give it to the least-loaded rank
            r = min(range(world), key=lambda q: (load[q], q))
            plan[r].append((e, v, n_videos))
            load[r] += (n_videos - v) * costs[e]
    return plan
