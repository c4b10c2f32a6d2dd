"""
parity.py -- full-depth, full-width comparison of the HIP visual path with the CPU oracle on SHARED weights.
*** TEST INFRASTRUCTURE ONLY *** (same rule as merv_oracle.py: tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline / parity leg may import it; merv_amd/ never does).

`reference_video(...)` pushes ONE video through oracle.merv_oracle with the very tensors the GPU path was built from
(merv/models/vidlms/merv.py:562-609: encoders -> [B,T,S,C] -> projectors -> fusion) and times it; `compare(...)`
reduces a (HIP, oracle) pair to the two numbers the tolerance is stated in: relative L2 over the whole tensor and the
minimum per-token cosine.
"""
from __future__ import annotations

import time
from typing import Dict, List, Sequence, Tuple

import torch

from . import merv_oracle as O


def spec_to_cfg(spec) -> O.EncoderCfg:
    """merv_amd.encoder.EncoderSpec -> oracle EncoderCfg (same field names by construction)."""
    return O.EncoderCfg(**{k: getattr(spec, k) for k in O.EncoderCfg.__dataclass_fields__})


def fusion_state(fusion_module) -> Dict[str, torch.Tensor]:
    """The reference state-dict entries oracle.fusion_forward reads (nn_utils.py:455-521), fp32 CPU."""
    sd = fusion_module.state_dict()
    return {k: sd[k].detach().float().cpu() for k in ("Q", "attention.q_proj_weight", "attention.k_proj_weight", "attention.in_proj_bias")}


def compare(hip: torch.Tensor, ref: torch.Tensor) -> Dict[str, float]:
    a = hip.detach().float().cpu().reshape(-1, hip.shape[-1])
    b = ref.detach().float().cpu().reshape(-1, ref.shape[-1])
    rel = float((a - b).norm() / (b.norm() + 1e-30))
    cos = float(torch.nn.functional.cosine_similarity(a, b, dim=-1).min())
    return {"rel_l2": round(rel, 6), "min_cos": round(cos, 6)}


@torch.no_grad()
def reference_video(pixels: Sequence[torch.Tensor], specs: Sequence, enc_W: Sequence[Dict],
                    proj_W: Sequence[Tuple[torch.Tensor, torch.Tensor]], Fw: Dict[str, torch.Tensor], out_size: int = 8):
    """pixels[i]: ONE video in encoder i's layout (CPU, any float dtype); enc_W / proj_W / Fw: fp32 CPU tensors, the same
    values the HIP path holds. Returns (result dict, seconds): tokens[i] [1, T*S, C], projected[i] [1, 1024, llm],
    fused [1, 1024, llm], weights [1, E]. Every consumed block is executed (23 / 23 / 12 / 11)."""
    t0 = time.perf_counter()
    tokens, projected = [], []
    for pix, spec, W, (pw, pb) in zip(pixels, specs, enc_W, proj_W):
        cfg = spec_to_cfg(spec)
        tok = O.encoder_forward(pix.float(), cfg, W)
        tokens.append(tok)
        projected.append(O.projector_forward(tok, cfg.t_out, cfg.hp, out_size, pw.float(), pb.float()))
    fused, w = O.fusion_forward(projected, Fw)
    return {"tokens": tokens, "projected": projected, "fused": fused, "weights": w}, time.perf_counter() - t0



@torch.no_grad()
def reference_video_workers(pixels: Sequence[torch.Tensor], specs: Sequence, enc_W: Sequence[Dict],
                            proj_W: Sequence[Tuple[torch.Tensor, torch.Tensor]], Fw: Dict[str, torch.Tensor], threads_per_worker: int,
                            out_size: int = 8, timeout: float = 900.0):
    """The same video as reference_video, with the E encoder branches (independent until fusion, merv.py:563-566) in E worker processes of
    `threads_per_worker` threads each (oracle/cpu_worker.py) -- the fair CPU baseline on a many-core host, where ONE torch intra-op pool
    stops scaling at ~16 threads on these shapes. Workers are child processes with fresh interpreters (the caller may hold a GPU context);
    their inputs travel as files in a temporary directory (/dev/shm when present), and they load them and build their thread pools BEFORE
    the clock starts. The clock covers release -> every branch's projected tokens written -> read back and fused in this process.
    Returns (fused, weights, seconds, per-worker compute seconds)."""
    import os
    import subprocess
    import sys
    import tempfile
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(prefix="merv_cpu_baseline_", dir=base) as tmp:
        procs = []
        try:
            for i, (pix, spec, W, (pw, pb)) in enumerate(zip(pixels, specs, enc_W, proj_W)):
                torch.save({"pix": pix.float(), "cfg": spec_to_cfg(spec), "W": W, "pw": pw.float(), "pb": pb.float(), "out_size": out_size},
                           f"{tmp}/in{i}.pt")
                env = dict(os.environ, OMP_NUM_THREADS=str(threads_per_worker), MKL_NUM_THREADS=str(threads_per_worker))
                procs.append(subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", f"{tmp}/in{i}.pt", f"{tmp}/out{i}.pt", str(threads_per_worker)],
                                              cwd=str(root), env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
            for p in procs:
                line = p.stdout.readline()
                if line.strip() != "ready":
                    raise RuntimeError(f"oracle.cpu_worker did not start: {line!r}")
            t0 = time.perf_counter()
            for p in procs:
                p.stdin.write("go\n")
                p.stdin.flush()
            per = []
            for p in procs:
                line = p.stdout.readline().split()
                if len(line) != 2 or line[0] != "done":
                    raise RuntimeError(f"oracle.cpu_worker failed: {line!r}")
                per.append(float(line[1]))
            projected = [torch.load(f"{tmp}/out{i}.pt") for i in range(len(procs))]
            fused, w = O.fusion_forward(projected, Fw)
            total = time.perf_counter() - t0
        finally:
            for p in procs:
                try:
                    p.stdin.close()
                    p.wait(timeout=20)
                except Exception:
                    p.kill()
    return fused, w, total, per
