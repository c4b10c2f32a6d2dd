"""
MervVisualPath: the visual branch of MERV.forward (merv/models/vidlms/merv.py:562-609) on MI355X --
E encoders -> reshape [B,T,S,C] -> 3davg+linear projectors -> cross-encoder fusion -> [B, 1024, llm] fused tokens.

This is the ONE orchestration of the path: `bench.py`, `MERVVisual.encode` / `MERV.generate` (merv_amd/vidlm.py), the
hipGraph capture and the multi-GPU placement (merv_amd/distributed.py) all run through it. Every buffer a step touches
(encoder workspace, encoder tokens, pooled rows, projected tokens, fusion partials / weights / output) is persistent
per (encoder, batch): a step performs no allocation and needs no cross-stream lifetime bookkeeping.

Single GPU: the encoders are independent until fusion (merv.py:563-566 is a list comprehension with no cross-talk),
so each runs on its own HIP stream, event-joined before the fusion kernels (the reference runs them one after the
other on one stream).

Multi GPU (torch.distributed over RCCL/xGMI, one process per GPU; merv_amd/distributed.py): work units
(encoder, video range, frame range) of a global batch are placed on ranks by `plan_units` (greedy contiguous fill over
their FLOP cost; every rank holds all encoder weights: 1.75 GB bf16 of 288 GB); each rank projects its units, ONE
collective (`all_to_all_single` by default, `all_gather_into_tensor` as the literal north_star form) moves the
projected [*, 64 rows per frame, llm] bf16 tokens to the rank that fuses the video.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import check, ptr
from .encoder import EncoderSpec, HipEncoder
from .projector import CrossAttentionAdapterLearnableQuery


# The encoder side streams are shared by every path object of a device. HIP multiplexes streams onto GPU_MAX_HW_QUEUES (default 4)
# hardware queues in creation order: the first four side streams of a process get one queue each, a second path's own four would be
# folded onto fewer queues and its encoders partly serialised (measured at one video per call: 11.4 against 9.4 ms; with
# GPU_MAX_HW_QUEUES=8 both 9.4; tools/probes/alloc_order_probe.py). Work on them is stream-ordered and event-joined per call, so
# paths that share them stay correct even when driven from different threads.
_ENCODER_STREAMS: Dict[int, List["torch.cuda.Stream"]] = {}


def _encoder_streams(device: torch.device, n: int) -> List["torch.cuda.Stream"]:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    pool = _ENCODER_STREAMS.setdefault(idx, [])
    while len(pool) < n:
        st = torch.cuda.Stream(device)
        # HIP binds a stream to its hardware queue at the stream's FIRST USE, not at creation: touch every side stream here, in creation order and from
        # this one thread, so that the four get one queue each whatever enqueues first later (round 6: with one host thread per chain the first uses
        # raced, and a process whose streams had landed badly ran its one-video step at 9.8 ms instead of 8.8 for good)
        with torch.cuda.device(device), torch.cuda.stream(st):
            torch.zeros(1, device=device)
        st.synchronize()
        pool.append(st)
    return pool[:n]


class MervVisualPath:
    def __init__(self, specs: Sequence[EncoderSpec], enc_weights: Optional[Sequence[Dict]],
                 proj_weights: Sequence[Tuple[torch.Tensor, torch.Tensor]],
                 fusion: Optional[CrossAttentionAdapterLearnableQuery], device, out_size: int = 8,
                 concurrent_streams: bool = True, encoders: Optional[Sequence[HipEncoder]] = None,
                 selectors: Optional[Sequence[Optional[Tuple]]] = None):
        """`encoders`: already-resident HipEncoder objects (then `enc_weights` is ignored), else one is built per
        (spec, weight dict). `fusion=None` is the single-encoder form (merv.py:607: no fusion module): forward returns
        the lone projector's output and weights None.
        `selectors[i]` = (fn, T, side) for a backbone whose forward() is NOT the plain patch selection (registry ids with a class-token /
        averaged / pooled selection, materialize.py:31-73): `fn(pixels) -> [B, T * side * side, C]` bf16 on the current stream replaces the
        encoder's patch-token output and (T, side) the projector's grid (merv.py:576-585: reshape to [B, temporal_resolution,
        spatial_resolution, C]; nn_utils.py:320-330 pools it to (T, 8, 8), replicating when side < 8). None: the spec's patches."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MervVisualPath needs a ROCm device; merv_amd has no CPU path")
        self.lib = _lib.load()
        self.specs = list(specs)
        with torch.cuda.device(self.device):
            if encoders is not None:
                self.encoders = list(encoders)
                for e in self.encoders:
                    if e.device != self.device:
                        raise ValueError(f"encoder {e.spec.name} lives on {e.device}, path on {self.device}")
            else:
                self.encoders = [HipEncoder(s, w, self.device) for s, w in zip(specs, enc_weights)]
            self.out_size = out_size
            self.set_parameters(proj_weights, fusion)
        self.selectors = list(selectors) if selectors is not None else [None] * len(self.specs)
        self.tokens_out = {(s.t_out if sel is None else sel[1]) * out_size * out_size for s, sel in zip(specs, self.selectors)}
        if len(self.tokens_out) != 1:  # merv.py:175-193 consistency assert
            raise ValueError(f"Output token length is not consistent across projectors: {self.tokens_out}")
        self.T_vis = self.tokens_out.pop()
        self.concurrent = concurrent_streams and len(self.encoders) > 1
        self.streams = _encoder_streams(self.device, len(self.encoders))
        # cost rank of every encoder (0 = most FLOPs per video): the stream map below is written in ranks
        order = sorted(range(len(self.specs)), key=lambda i: -self.specs[i].flops_per_video())
        self._rank = {i: r for r, i in enumerate(order)}
        # the chain that ends a concurrent step -- the largest encoder's -- takes the fast wide form for its sub-round GEMM launches, the others keep
        # the narrow one that leaves it the CUs (round 6: one video 9.43 -> 8.9 ms; launches of a full round or more -- >= 4 videos -- do not depend on it)
        if concurrent_streams and len(self.encoders) > 1:
            for i, enc in enumerate(self.encoders):
                if hasattr(enc, "set_latency_critical"):
                    enc.set_latency_critical(self._rank[i] == 0)
        self._stream_map_env = _lib.tuning("MERV_ENCODER_STREAM_MAP")  # probe hook, by rank: "0123" = one stream per encoder, "0111", ...
        self.threaded_enqueue: Optional[bool] = None  # see _threaded_enqueue
        self._executor, self._executor_pid = None, -1
        self._bufs: Dict[Tuple[int, int, int], Dict[str, torch.Tensor]] = {}
        self._fuse_bufs: Dict[int, Dict[str, torch.Tensor]] = {}
        self._events: Dict[int, Tuple[torch.cuda.Event, List[torch.cuda.Event]]] = {}

    def set_parameters(self, proj_weights: Sequence[Tuple[torch.Tensor, torch.Tensor]],
                       fusion: Optional[CrossAttentionAdapterLearnableQuery]) -> None:
        """(Re)take the trainable tail's parameters: bf16 device copies of the projector weights, fp32 biases, and the
        fusion module's folded query vector. The frozen encoders and every buffer stay as they are."""
        if fusion is None and len(self.encoders) != 1:
            raise TypeError("argument of type 'NoneType' is not iterable")  # reference behaviour, merv.py:607 (App. B.4)
        with torch.cuda.device(self.device):
            old = getattr(self, "proj", None)
            if old is not None and all(o[0].shape == w.shape for o, (w, _) in zip(old, proj_weights)):
                for (ow, ob), (w, b) in zip(old, proj_weights):  # in place, ordered on the current stream behind the last
                    ow.copy_(w.detach())                          # step's event join: no buffer a side stream reads is freed
                    ob.copy_(b.detach())
            else:
                self.proj = [(w.detach().to(self.device, torch.bfloat16).contiguous(),
                              b.detach().to(self.device, torch.float32).contiguous()) for w, b in proj_weights]
            self.llm_dim = self.proj[0][0].shape[0]
            self.fusion = fusion
            if fusion is not None:
                self.fusion.prepare(self.device)

    # -- buffers are persistent per (encoder, batch, frames): no allocator traffic and no cross-stream lifetime issues
    def _enc_bufs(self, i: int, B: int, frames: Optional[int] = None) -> Dict[str, torch.Tensor]:
        s = self.specs[i]
        t_out = s.t_out if frames is None else frames // s.tubelet
        s_out = s.s_out
        if self.selectors[i] is not None:
            if frames is not None:
                raise ValueError(f"{s.name}: frame-range units need the plain patch selection")
            t_out, s_out = self.selectors[i][1], self.selectors[i][2] ** 2
        key = (i, B, t_out)
        if key not in self._bufs:
            o = self.out_size
            self._bufs[key] = {
                "tokens": torch.empty(B, t_out * s_out, s.dim, dtype=torch.bfloat16, device=self.device),
                "pooled": torch.empty(B * t_out * o * o, s.dim, dtype=torch.bfloat16, device=self.device),
                "proj": torch.empty(B, t_out * o * o, self.llm_dim, dtype=torch.bfloat16, device=self.device),
            }
            self.encoders[i].workspace(B)
        return self._bufs[key]

    def buffers(self, i: int, B: int, frames: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """Encoder i's persistent buffers for batch B: "tokens" [B, T*S, C] (a4-a8 output), "pooled", "proj" [B, T*64, llm]
        (a9 output) -- what the last forward of that shape left there (inspection / parity checks)."""
        return self._enc_bufs(i, B, frames)

    def encode_project(self, i: int, pixels: torch.Tensor, stream: Optional[torch.cuda.Stream] = None,
                       out: Optional[torch.Tensor] = None, frames: Optional[int] = None) -> torch.Tensor:
        """a4-a9 for encoder i on `stream`: pixels -> projected [B, T_vis, llm] bf16. `frames` (per-frame encoders only):
        the pixel tensor holds that many frames per video instead of the spec's count (a frame-range work unit of the
        multi-GPU placement; LanguageBind in whole clips of 8): the result is the matching [B, frames/tubelet*64, llm] row
        range of the full projection, bit for bit (frames are independent sequences and the pool does not cross frames)."""
        s = self.specs[i]
        B = pixels.shape[0]
        if frames is not None and frames == s.frames:
            frames = None
        with torch.cuda.device(self.device):
            if stream is None:
                stream = torch.cuda.current_stream(self.device)
            bufs = self._enc_bufs(i, B, frames)
            t_out, side = (s.t_out if frames is None else frames // s.tubelet), s.hp
            if self.selectors[i] is None:
                tok = self.encoders[i].forward(pixels, out=bufs["tokens"], stream=stream, frames=frames)
            else:  # the backbone's own forward() (encoder + token selection) on this branch's stream, into the persistent buffer
                fn, t_out, side = self.selectors[i]
                with torch.cuda.stream(stream):
                    sel = fn(pixels)
                    if sel.numel() != bufs["tokens"].numel():  # the reference's reshape(-1, T, S, C) (merv.py:576-585) would fail too
                        raise RuntimeError(f"{s.name}: forward() returned {tuple(sel.shape)}, which is not [B, {t_out} * {side * side}, {s.dim}]")
                    tok = bufs["tokens"].copy_(sel.reshape(bufs["tokens"].shape))
            dst = out if out is not None else bufs["proj"]
            w, b = self.proj[i]
            rc = self.lib.merv_projector_forward(ptr(tok), B, t_out, side, s.dim, self.out_size, ptr(w), ptr(b),
                                                 self.llm_dim, ptr(bufs["pooled"]), ptr(dst), stream.cuda_stream)
            check(rc, "merv_projector_forward")
        return dst

    def fuse(self, projected: Sequence[torch.Tensor]):
        """a10 on the current stream into persistent buffers. Returns (fused [B,T_vis,llm] bf16, weights [B,E] fp32)."""
        if self.fusion is None:
            return projected[0], None
        B = projected[0].shape[0]
        fb = self._fuse_bufs.get(B)
        if fb is None:
            E = len(projected)
            fb = self._fuse_bufs[B] = {
                "partial": torch.empty(self.lib.merv_fusion_workspace_floats(B, E, self.T_vis), dtype=torch.float32, device=self.device),
                "weights": torch.empty(B, E, dtype=torch.float32, device=self.device),
                "out": torch.empty(B, self.T_vis, self.llm_dim, dtype=torch.bfloat16, device=self.device),
            }
        with torch.cuda.device(self.device):
            return self.fusion(list(projected), out=fb["out"], partial=fb["partial"], weights=fb["weights"])

    def stream_map(self, batch: int) -> List[int]:
        """Side stream of every encoder at this batch size. More concurrent kernel chains fill the tails of each other's launches, fewer
        contend less for the per-XCD L2s (every chain streams its own weight panel and activations through them). Measured on the
        merv-full four (round 5, EXPERIMENTS.md section 4; ms per step, one stream per encoder -> the map): 16 videos 114.4-117.2 -> 113.1-115.7
        with the largest encoder alone and the other three back to back on ONE stream; 8 / 4 / 2 videos 59.8 / 31.0 / 17.1 -> 58.9 / 30.5 / 16.7
        with the largest alone, the two middle ones sharing, the smallest alone; at one video every encoder keeps its own stream (10.0 ms
        against 10.3 / 11.5: small launches need the company). Other encoder counts: one stream each."""
        E = len(self.encoders)
        m = self._stream_map_env
        if m and not (len(m) == E and all(c.isdigit() and int(c) < len(self.streams) for c in m)):
            raise ValueError(f"MERV_ENCODER_STREAM_MAP={m!r}: expected {E} digits, each below {len(self.streams)} (a side stream per encoder rank)")
        if not m:
            m = ("0111" if batch >= 12 else "0112" if batch >= 2 else "0123") if E == 4 else "".join(str(i) for i in range(E))
        return [int(m[self._rank[i]]) for i in range(E)]

    def enqueue_order(self, batch: int) -> List[int]:
        """Order in which the branches are enqueued (= executed, for encoders that share a stream): the largest encoder first (it has a
        stream of its own), then the others from the SMALLEST up, so that the chain which ends the step alone is the second-largest encoder's
        -- full-chip launches -- and the small encoders' under-filled launches run beside the largest's (16 videos, map 0111: 113.2-113.4 ms
        with ranks 0 1 2 3, 112.6-112.9 with 0 3 2 1 or 0 2 3 1, 113.5-113.6 with 0 1 3 2 / 0 3 1 2). With a stream per encoder (one video: what
        every generate() call is) the order only decides when the host gets round to each chain -- one encoder's launches take the host 0.3-1.2 ms
        to enqueue --, so the chains start longest first: a single call 10.2 -> 9.35 ms (median of 20, round 6: tools/sessions/gpu_r6_s8.sh; the
        back-to-back rate, 9.5 ms, does not care). Probe hook MERV_ENCODER_ORDER: ranks."""
        E = len(self.encoders)
        o = _lib.tuning("MERV_ENCODER_ORDER")
        by_rank = sorted(range(E), key=lambda i: self._rank[i])
        if o and sorted(o) == [str(r) for r in range(E)]:
            return [by_rank[int(c)] for c in o]
        if len(set(self.stream_map(batch))) == E:
            return by_rank
        return by_rank[:1] + by_rank[:0:-1]

    def _threaded_enqueue(self, smap: Sequence[int]) -> bool:
        """Enqueue the branches from one host thread each? When every encoder has a stream of its own (one video per call: what generate() makes),
        never inside a graph capture (a capture belongs to the capturing thread). A step is ~700 launches = 2.0 ms of host time on one thread, so
        the last chain starts 1-2 ms after the first; with the sub-round policy of round 6 the GPU needs 8.7-9.0 ms, and a single call takes
        9.16-9.23 ms enqueued by one thread, 8.70-8.72 by four (tools/probes/threaded_enqueue_probe.py, same bits; the back-to-back rate, where the
        host runs ahead anyway, is the same). `threaded_enqueue` (None = that rule, False = never) overrides."""
        if torch.cuda.is_current_stream_capturing():
            return False
        if self.threaded_enqueue is not None:
            return bool(self.threaded_enqueue) and len(set(smap)) == len(smap)
        return _lib.tuning("MERV_THREADED_ENQUEUE", "1") != "0" and len(set(smap)) == len(smap)

    def _pool(self):
        import os
        if self._executor is None or self._executor_pid != os.getpid():  # (threads do not survive a fork: a child process makes its own pool)
            from concurrent.futures import ThreadPoolExecutor
            self._executor = ThreadPoolExecutor(max_workers=max(1, len(self.encoders) - 1), thread_name_prefix="merv-enqueue")
            self._executor_pid = os.getpid()
        return self._executor

    def _run_branches(self, pixels: Sequence[torch.Tensor], project: bool) -> List[torch.Tensor]:
        """The E independent branches (merv.py:563-566), each on its own stream when `concurrent`, event-joined on the
        current stream. project=True: a4-a9 (projected tokens); False: a4-a8 only (encoder tokens [B, T*S, C])."""
        if len(pixels) != len(self.encoders):
            raise ValueError(f"expected {len(self.encoders)} pixel tensors, got {len(pixels)}")
        main = torch.cuda.current_stream(self.device)

        def branch(i, pix, st):
            if project:
                return self.encode_project(i, pix, st)
            if self.selectors[i] is not None:
                raise NotImplementedError(f"{self.specs[i].name}: encode_tokens() (the training step) drives plain patch selections only")
            bufs = self._enc_bufs(i, pix.shape[0])
            return self.encoders[i].forward(pix, out=bufs["tokens"], stream=st)

        outs = []
        if self.concurrent:
            if torch.cuda.is_current_stream_capturing():  # events recorded inside a capture belong to that graph
                start, dones = torch.cuda.Event(), [torch.cuda.Event() for _ in pixels]
            else:
                if 0 not in self._events:
                    self._events[0] = (torch.cuda.Event(), [torch.cuda.Event() for _ in pixels])
                start, dones = self._events[0]
            start.record(main)
            smap = self.stream_map(pixels[0].shape[0])
            outs = [None] * len(pixels)
            order = self.enqueue_order(pixels[0].shape[0])
            if self._threaded_enqueue(smap):
                # a stream per encoder (one video per call): every chain is enqueued by its own host thread (an encoder is ONE library call,
                # which runs without the GIL), so the chains start together instead of 0.3-1.2 ms apart -- the host needs ~2 ms to enqueue a
                # step's ~700 launches -- and the largest encoder's chain, which ends the step, is never waiting for the host
                inference, grad = torch.is_inference_mode_enabled(), torch.is_grad_enabled()  # (thread-local modes: the workers take the caller's)

                def work(i):
                    with torch.cuda.device(self.device), torch.inference_mode(inference), torch.set_grad_enabled(grad):
                        st = self.streams[smap[i]]
                        st.wait_event(start)
                        outs[i] = branch(i, pixels[i], st)
                        dones[i].record(st)
                futs = [self._pool().submit(work, i) for i in order[1:]]
                work(order[0])
                for f in futs:
                    f.result()
                for i in order:
                    main.wait_event(dones[i])
            else:
                for i in order:  # encoders that share a stream run in this order
                    st = self.streams[smap[i]]
                    st.wait_event(start)
                    outs[i] = branch(i, pixels[i], st)
                    dones[i].record(st)
                    main.wait_event(dones[i])
        else:
            for i, pix in enumerate(pixels):
                outs.append(branch(i, pix, main))
        return outs

    def encode_tokens(self, pixels: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """a4-a8: the encoders' patch tokens [B, T*S, C] bf16 (persistent buffers), for callers that differentiate the
        projector / fusion themselves (merv_amd/train.py)."""
        with torch.cuda.device(self.device):
            return self._run_branches(pixels, project=False)

    def forward(self, pixels: Sequence[torch.Tensor]):
        """pixels[i]: encoder i's post-transform tensor. Returns (fused [B,T_vis,llm] bf16, weights [B,E] fp32 | None).
        The returned tensors are the path's persistent buffers: the next forward of the same batch size overwrites them
        (stream-ordered), so clone what must outlive it."""
        with torch.cuda.device(self.device):
            return self.fuse(self._run_branches(pixels, project=True))

    def capture(self, pixels: Sequence[torch.Tensor]):
        """Record one forward for these input shapes into a hipGraph (torch.cuda.CUDAGraph: the library's launches go
        to torch's capturing stream, the per-encoder side streams fork from and re-join it through events, so the four
        encoder chains stay concurrent inside the graph). Returns `replay(new_pixels=None) -> (fused, weights)`; the
        outputs are the same tensors on every replay. At batch 1 a step is ~700 launches of 10-40 us kernels, so the
        graph removes most of the launch overhead; at batch 8 it is within noise."""
        self.forward(pixels)  # warm-up outside the capture: kernel attributes, workspaces, persistent buffers
        torch.cuda.synchronize(self.device)
        with torch.inference_mode(False):  # a capture made under inference_mode must stay writable by later, ordinary callers
            static = [torch.empty(p.shape, dtype=p.dtype, device=p.device) for p in pixels]
        for dst, src in zip(static, pixels):
            dst.copy_(src)
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
def plan_units(costs: Sequence[float], n_videos: int, world: int, frames: Optional[Sequence[int]] = None,
               atoms: Optional[Sequence[int]] = None) -> List[List[Tuple[int, int, int, int, int]]]:
    """Place the visual path's independent work on `world` ranks. Returns, per rank, units (e, v0, v1, f0, f1): encoder e
    on videos [v0, v1) and frames [f0, f1) of each (v1 - v0 > 1 only with whole videos).

    costs[e] = FLOPs of encoder e per video; frames[e] = frames per video; atoms[e] = the smallest frame count that is an
    independent piece of work for encoder e: 1 for per-frame encoders (DINOv2, SigLIP: every frame is
    its own sequence, dinov2_video.py:135-136), the temporal-attention clip length for LanguageBind (8: clips are independent
    by construction, modeling_video.py:140-146), all frames for a joint space-time encoder (ViViT). Without frames / atoms
    only whole videos are placed.

    Method: the atoms are laid on a line ordered by (encoder cost descending, video, frame) and the line is cut into at most
    `world` contiguous segments of minimal maximum cost (exact for this order: bisection on the bound + greedy fill), so a
    rank holds, per encoder, ONE contiguous run of atoms = at most a partial leading video, a batch of whole videos and a
    partial trailing video. With 8 videos per rank the imbalance is < 1 %; with ONE video on 4 / 8 ranks the makespan is
    1.89 / 1.64 TFLOP against 3.28 for one-encoder-per-GPU. Deterministic; every (e, v, frame) appears exactly once."""
    E = len(costs)
    frames = list(frames) if frames is not None else [1] * E
    atoms = list(atoms) if atoms is not None else list(frames)
    for e in range(E):
        if atoms[e] <= 0 or frames[e] % atoms[e]:
            raise ValueError(f"encoder {e}: atom of {atoms[e]} frames does not divide {frames[e]}")
    order = sorted(range(E), key=lambda e: -costs[e])
    line: List[Tuple[int, int, int, float]] = []  # (e, v, atom index, cost)
    for e in order:
        per = frames[e] // atoms[e]
        for v in range(n_videos):
            for a in range(per):
                line.append((e, v, a, costs[e] / per))
    if not line:
        return [[] for _ in range(world)]

    def cut(bound: float) -> Optional[List[int]]:
        """Greedy fill under `bound`: segment start indices, or None if more than `world` segments are needed."""
        starts, load = [0], 0.0
        for i, (_, _, _, c) in enumerate(line):
            if load + c > bound * (1 + 1e-12) and load > 0.0:
                starts.append(i)
                load = 0.0
                if len(starts) > world:
                    return None
            load += c
        return starts

    lo = max(max(c for *_, c in line), sum(c for *_, c in line) / world)
    hi = sum(c for *_, c in line)
    best = cut(hi)
    if cut(lo) is not None:
        best = cut(lo)
    else:
        for _ in range(60):
            mid = 0.5 * (lo + hi)
            got = cut(mid)
            if got is None:
                lo = mid
            else:
                hi, best = mid, got
    bounds = best + [len(line)]
    plan: List[List[Tuple[int, int, int, int, int]]] = [[] for _ in range(world)]
    for r in range(len(best)):
        seg = line[bounds[r]:bounds[r + 1]]
        i = 0
        while i < len(seg):  # group the segment's atoms of one encoder into (partial video | whole videos | partial video)
            e, v, a, _ = seg[i]
            per = frames[e] // atoms[e]
            j = i
            while j < len(seg) and seg[j][0] == e and seg[j][1] == v:
                j += 1
            if a == 0 and j - i == per:  # whole video: extend over following whole videos of the same encoder
                v1 = v + 1
                while j + per <= len(seg) and seg[j][0] == e and seg[j][1] == v1 and seg[j][2] == 0 and \
                        seg[j + per - 1][0] == e and seg[j + per - 1][1] == v1:
                    j += per
                    v1 += 1
                plan[r].append((e, v, v1, 0, frames[e]))
            else:
                plan[r].append((e, v, v + 1, a * atoms[e], (a + (j - i)) * atoms[e]))
            i = j
    return plan


def plan_one_encoder_per_rank(n_encoders: int, n_videos: int, world: int, frames: Sequence[int]):
    """The literal BASELINE.json configs[2] placement: encoder e runs on rank e % world for every video."""
    plan: List[List[Tuple[int, int, int, int, int]]] = [[] for _ in range(world)]
    for e in range(n_encoders):
        plan[e % world].append((e, 0, n_videos, 0, frames[e]))
    return plan
