#!/usr/bin/env python3
"""
bench.py -- fused visual tokens/s through the 4-encoder + projector + fusion path (merv-frozen / merv-full geometry,
frames [16,16,32,16], bf16) on N MI355X. Contract: see the task statement; ONE JSON line on rank 0.

  step      = one pass of the hot path (a4-a10: pixels resident in HBM -> fused [B,1024,4096] bf16) over one batch
  value     = fused visual tokens / s, whole job (1024 tokens per video)
  roofline  = the bf16 MFMA GEMM kernel (92 % of the path's FLOPs): algorithmic FLOPs (2MNK) of all its launches in K
              steps / their summed HIP-event durations (events recorded by the library on the launch stream)
  cpu_baseline = the CPU oracle (torch fp32) timed on this host on a bounded sample, extrapolated per layer
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP8_TFLOPS = 5000.0  # dense MX-scaled fp8 (MI355X_MICROARCH.md, Matrix cores)
LLM_DIM, FUSION_EMBED = 4096, 3072
TOKENS_PER_VIDEO = 1024


def device_random_weights(spec, seed, device):
    """Seeded random-init weights of the named architecture, generated directly on the GPU (no checkpoints here)."""
    g = torch.Generator(device=device).manual_seed(seed)
    D, Mh = spec.dim, spec.mlp_dim

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g, device=device) * std

    P = spec.s_out * (spec.t_out if spec.joint_space_time else 1)
    W = {"patch_w": rn(D, spec.k_true, std=spec.k_true**-0.5), "pos": rn(P, D), "layers": []}
    if spec.name != "languagebind":
        W["patch_b"] = rn(D)
    if spec.prefix_tokens:
        W["prefix"] = rn(spec.prefix_tokens, D)
    if spec.pre_ln:
        W["pre_ln_w"], W["pre_ln_b"] = 1 + rn(D, std=0.1), rn(D, std=0.1)
    if spec.final_ln:
        W["final_ln_w"], W["final_ln_b"] = 1 + rn(D, std=0.1), rn(D, std=0.1)
    for _ in range(spec.layers):
        Lw = {"ln1_w": 1 + rn(D, std=0.1), "ln1_b": rn(D, std=0.1), "qkv_w": rn(3 * D, D, std=D**-0.5), "qkv_b": rn(3 * D),
              "proj_w": rn(D, D, std=D**-0.5), "proj_b": rn(D), "ln2_w": 1 + rn(D, std=0.1), "ln2_b": rn(D, std=0.1),
              "fc1_w": rn(Mh, D, std=D**-0.5), "fc1_b": rn(Mh), "fc2_w": rn(D, Mh, std=Mh**-0.5), "fc2_b": rn(D)}
        if spec.layerscale:
            Lw["ls1"], Lw["ls2"] = 0.5 + rn(D, std=0.2), 0.5 + rn(D, std=0.2)
        if spec.temporal_frames:
            Lw.update({"t_emb": rn(spec.temporal_frames, D, std=D**-0.5), "t_ln_w": 1 + rn(D, std=0.1),
                       "t_ln_b": rn(D, std=0.1), "t_qkv_w": rn(3 * D, D, std=D**-0.5), "t_qkv_b": rn(3 * D),
                       "t_proj_w": rn(D, D, std=D**-0.5), "t_proj_b": rn(D)})
        W["layers"].append(Lw)
    return W


def build_path(device, concurrent=True):
    from merv_amd.encoder import merv_full_specs
    from merv_amd.projector import CrossAttentionAdapterLearnableQuery
    from merv_amd.visual_path import MervVisualPath
    specs = merv_full_specs()
    enc_w = [device_random_weights(s, 1000 + i, device) for i, s in enumerate(specs)]
    g = torch.Generator(device=device).manual_seed(77)
    proj_w = [(torch.randn(LLM_DIM, s.dim, generator=g, device=device) * s.dim**-0.5,
               torch.randn(LLM_DIM, generator=g, device=device) * 0.02) for s in specs]
    torch.manual_seed(1024)  # merv.py:87
    fusion = CrossAttentionAdapterLearnableQuery(embed_dim=FUSION_EMBED, llm_dim=LLM_DIM, token_length=TOKENS_PER_VIDEO,
                                                 averagetoken=True)
    path = MervVisualPath(specs, enc_w, proj_w, fusion, device, concurrent_streams=concurrent)
    del enc_w
    return specs, path


def synth_pixels(specs, n_videos, device, seed):
    """Post-transform pixel tensors (~N(0,1) after normalisation), bf16, already resident in HBM."""
    g = torch.Generator(device=device).manual_seed(seed)
    return [torch.randn(s.pixel_shape(n_videos), generator=g, device=device).to(torch.bfloat16) for s in specs]


def cpu_baseline(budget_layers=2, threads=None):
    """Oracle (torch fp32) on ONE video through the whole path: patch embed, every consumed block of the four encoders
    (weights for `budget_layers` blocks are generated and reused cyclically: timing does not depend on their values),
    projectors and fusion -- about 10-15 s of CPU work on the GPU host. Thread count: torch's intra-op pool
    degrades badly past ~16 threads on these shapes (measured on the 256-core GPU host: 16 threads 0.118 s, 64 threads
    0.30 s, 256 threads 6.5 s for the same two SigLIP blocks), so the baseline uses min(cores, 16) and says so."""
    from oracle import merv_oracle as O
    ncores = threads or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(ncores)
    cfgs = O.merv_full_cfgs()
    total = 0.0
    projected = []
    executed = skipped = 0
    for i, cfg in enumerate(cfgs):
        depth = cfg.layers
        cfg_s = O.EncoderCfg(**{**cfg.__dict__, "layers": budget_layers})
        W = O.random_encoder_weights(cfg_s, seed=i)
        shape = (1, 3, cfg.frames, cfg.img, cfg.img) if cfg.pix_layout == "BCFHW" else (1, cfg.frames, 3, cfg.img, cfg.img)
        pix = torch.randn(shape, generator=torch.Generator().manual_seed(i))
        with torch.no_grad():
            t0 = time.perf_counter()
            x = O.encoder_embed(pix, cfg_s, W)
            t1 = time.perf_counter()
            # every consumed block is executed (the timing does not depend on the weight values, so the generated
            # blocks are reused cyclically); only if the host is so slow that the sample would pass ~45 s does the rest
            # of an encoder fall back to per-block extrapolation, and the sample string says so
            done = 0
            for li in range(depth):
                x = O.encoder_block(x, cfg_s, W["layers"][li % budget_layers])
                done += 1
                if total + (time.perf_counter() - t0) > 45.0 and done >= budget_layers:
                    break
            t2 = time.perf_counter()
            executed += done
            skipped += depth - done
            tok = x[:, cfg.prefix_tokens:].reshape(1, -1, cfg.dim)
            pw, pb = O.random_projector_weights(cfg.dim, LLM_DIM, seed=i)
            t3 = time.perf_counter()
            projected.append(O.projector_forward(tok, cfg.t_out, cfg.hp, 8, pw, pb))
            t4 = time.perf_counter()
        total += (t1 - t0) + (t2 - t1) / done * depth + (t4 - t3)
    Fw = O.random_fusion_weights(LLM_DIM, FUSION_EMBED, seed=5)
    with torch.no_grad():
        t0 = time.perf_counter()
        O.fusion_forward(projected, Fw)
        total += time.perf_counter() - t0
    return {
        "value": round(TOKENS_PER_VIDEO / total, 2), "unit": "visual-tokens/s", "cores": ncores, "kind": "port",
        "sample": (f"1 video through the whole path, fp32 torch CPU oracle: patch embed, all consumed blocks (23/23/12/11; "
                   f"{executed} executed" + (f", {skipped} extrapolated per block" if skipped else "") +
                   f"), projectors and fusion; {total:.1f} s/video"),
    }


def main():
    # Native libraries (RCCL's version banner, rocm warnings) print to fd 1; the contract is ONE JSON line on stdout, so
    # everything else is sent to stderr and the JSON goes to a private copy of the original stdout.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="videos per GPU per step")
    ap.add_argument("--sequential", action="store_true", help="run the encoders on one stream (reference behaviour)")
    ap.add_argument("--parallelism", default="dp", choices=["dp", "units"],
                    help="N > 1: 'dp' = every rank runs the whole path on its own videos (independent units, no data-path "
                         "collective; the weak-scaling form of configs[1]); 'units' = (encoder, video) units placed across "
                         "ranks with an RCCL exchange of projected tokens before fusion (configs[2]'s encoder sharding)")
    ap.add_argument("--exchange", default="all_to_all", choices=["all_to_all", "all_gather"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ln-fold", action="store_true", help="fold LN1 / LN2 into the qkv / fc1 GEMMs (exact algebra; opt-in)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph (single GPU; useful at --batch 1)")
    ap.add_argument("--mxfp8", action="store_true",
                    help="BASELINE.json configs[4] variant: block GEMMs on MXFP8 operands (NOT the headline bf16 metric; "
                         "dtype is reported as mxfp8 and the roofline peak as the dense fp8 peak)")
    ap.add_argument("--no-prof", action="store_true", help="skip the roofline leg (per-launch GEMM events, N=1 only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the HIP path has no CPU fallback")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from merv_amd import _lib
    lib = _lib.load()
    if os.environ.get("MERV_GEMM_GROUP_M"):  # tuning hook: tile-order group size of every GEMM launch
        lib.merv_debug_set_gemm_variant(int(os.environ["MERV_GEMM_GROUP_M"]) << 8)
    specs, path = build_path(device, concurrent=not args.sequential)
    if args.ln_fold:
        for enc in path.encoders:
            enc.enable_ln_fold()
    if args.mxfp8:
        for enc in path.encoders:
            enc.enable_mxfp8()
    B = args.batch
    G = B * world

    force_dist = os.environ.get("MERV_BENCH_FORCE_DISTRIBUTED") == "1"  # exercise the N>1 code path on one GPU
    if world == 1 and force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    use_units = force_dist or (world > 1 and args.parallelism == "units")
    if not use_units:
        pixels = synth_pixels(specs, B, device, seed=rank)  # this rank's own videos
        replay = path.capture(pixels) if args.graph else None

        def step():
            return replay() if (replay is not None and path.concurrent) else path.forward(pixels)
    else:
        from merv_amd.distributed import DistributedVisualPath
        dpath = DistributedVisualPath(path, [s.flops_per_video() for s in specs], world, rank, B, exchange=args.exchange)
        unit_pixels = dpath.synth_unit_pixels(specs, seed=1234)

        def step():
            return dpath.forward(unit_pixels)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out[0].float()).all()

    # ---- roofline leg: the same K steps again with every GEMM launch bracketed by HIP events on its own stream.
    # Kernel durations are only well defined when kernels do not overlap, so this pass runs the encoders on ONE
    # stream (the throughput above is measured with concurrent streams and no events).
    roof = None
    if not args.no_prof and world == 1 and not force_dist:
        was = path.concurrent
        path.concurrent = False
        step(); torch.cuda.synchronize()
        lib.merv_prof_reset()
        lib.merv_prof_enable(1)  # class 0: GEMM
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        lib.merv_prof_enable(0)
        path.concurrent = was
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        lib.merv_prof_read(0, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by))
        if n.value:
            achieved = fl.value / (ms.value * 1e-3) / 1e12
            traffic = None
            tf = ROOT / "profiles" / "r01_pmc_gemm_traffic.json"
            if tf.exists():
                try:
                    per_step = json.loads(tf.read_text()).get("hbm_bytes_per_step")
                    traffic = per_step / (n.value / args.steps) if per_step else None  # per GEMM call, like `achieved`
                except Exception:
                    traffic = None
            roof = {"bound": "mfma", "kernel": "gemm_bf16_8phase_kernel + gemm_bf16_kernel for the remaining rows (every GEMM call of the step)",
                    "achieved": round(achieved, 1), "peak": PEAK_FP8_TFLOPS if args.mxfp8 else PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / (PEAK_FP8_TFLOPS if args.mxfp8 else PEAK_BF16_TFLOPS), 4),
                    "traffic": None if args.mxfp8 else traffic,
                    "launches": n.value, "avg_launch_us": round(ms.value * 1e3 / n.value, 2),
                    "flops_per_launch": round(fl.value / n.value / 1e9, 3), "flops_unit": "GFLOP",
                    "algorithmic_bytes_per_launch": round(by.value / n.value),
                    "gemm_ms_per_step": round(ms.value / args.steps, 3)}
        lib.merv_prof_reset()
    if world > 1:
        dist.barrier()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = G * TOKENS_PER_VIDEO * args.steps / elapsed
        flops_video = sum(s.flops_per_video() + 2.0 * TOKENS_PER_VIDEO * s.dim * LLM_DIM for s in specs)  # + projectors
        path_tflops = flops_video * G * args.steps / elapsed / 1e12
        line = {
            "metric": "fused visual tokens/s through 4-encoder+projector+fusion (merv-full geometry)",
            "value": round(value, 1), "unit": "visual-tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "mxfp8 block GEMMs, bf16 elsewhere" if args.mxfp8 else "bf16", "data": "synthetic",
            "config": {"workload": ("merv-full encoders with MXFP8 block GEMMs (BASELINE.json configs[4] encoder side), "
                                    "frames [16,16,32,16], 224px" if args.mxfp8 else
                                    "merv-frozen 4 frozen encoders bf16 inference, frames [16,16,32,16], 224px "
                                    "(BASELINE.json configs[1])"),
                       "videos_per_gpu_per_step": B, "global_videos_per_step": G, "tokens_per_video": TOKENS_PER_VIDEO,
                       "encoder_streams": ("sequential" if args.sequential else "concurrent") + (", hipGraph replay" if args.graph else ""),
                       "parallelism": ("single GPU" if world == 1 and not force_dist else
                                       f"(encoder,video) units over {world} GPUs, RCCL {args.exchange} before fusion" if use_units else
                                       f"data-parallel over videos on {world} GPUs, no data-path collective"),
                       "path_tflops": round(path_tflops, 1),
                       "path_frac_of_mfma_peak": round(path_tflops / (PEAK_FP8_TFLOPS if args.mxfp8 else PEAK_BF16_TFLOPS), 4),
                       "flops_per_video_T": round(flops_video / 1e12, 3)},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), file=real_stdout, flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
