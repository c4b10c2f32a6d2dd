#!/usr/bin/env python3
"""One-GPU timing of the BASELINE.json configs[3] step: merv-full, projector + fusion + LLM unfrozen ("finetune" stage),
Llama-2-7B geometry with fp32 master weights under bf16 autocast, synthetic video batches, per-device batch 8
(conf/models.py:139 finetune_per_device_batch_size). Random-init weights (no checkpoints here). Prints one JSON line
and writes it to gpurun_out/train_bench.json. Under torchrun the same script runs data-parallel over RCCL."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--text-len", type=int, default=64)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--stage", default="finetune")
    ap.add_argument("--llm-layers", type=int, default=32)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    from merv_amd.backbones import get_video_backbone_and_transform
    from merv_amd.llm import LlamaBackbone, llama2_7b_config
    from merv_amd.registry import MODEL_CONFIGS
    from merv_amd.train import TrainStep
    from merv_amd.vidlm import MERV
    cfg = MODEL_CONFIGS["merv-full"]
    bbs, _ = get_video_backbone_and_transform(cfg["video_backbone_ids"], cfg["image_resize_strategy"], cfg["num_frames"],
                                              weights=["random"] * 4, device=dev)
    llm = LlamaBackbone(dict(llama2_7b_config(), num_hidden_layers=a.llm_layers), device=dev, dtype=torch.float32)
    m = MERV(bbs, llm, arch_specifier=cfg["arch_specifier"], feature_fusion=cfg["feature_fusion"],
             projector_token_length=cfg["projector_token_length"], visual_feature_length=cfg["visual_feature_length"]).to(dev)
    ts = TrainStep(m, stage=a.stage, learning_rate=2e-5, weight_decay=0.1, max_grad_norm=1.0, warmup_ratio=0.03, max_steps=1000)
    g = torch.Generator().manual_seed(rank)
    B, S = a.batch, a.text_len
    ids = torch.randint(3, 32000, (B, S), generator=g)
    ids[:, 0] = 1
    labels = ids.clone()
    labels[:, : S // 2] = -100
    batch = dict(input_ids=ids.to(dev), attention_mask=torch.ones(B, S, dtype=torch.bool, device=dev), labels=labels.to(dev),
                 video_values=[torch.randn(B, *b.default_video_resolution, generator=g).to(dev) for b in bbs],
                 multimodal_indices=torch.arange(B, device=dev))
    for _ in range(a.warmup):
        info = ts.step(batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        info = ts.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank == 0:
        n_train = sum(p.numel() for p in ts.sync.params)
        out = {"workload": f"merv-full {a.stage} step, Llama-2-7B geometry ({a.llm_layers} layers), frames [16,16,32,16], "
                           f"per-device batch {B}, text {S} + 1024 visual tokens", "n_gpus": world, "s_per_step": dt,
               "samples_per_s": B * world / dt, "llm_tokens_per_s": B * world * (S + 1024) / dt, "trainable_params": n_train,
               "loss": info["loss"], "grad_norm": info["grad_norm"], "peak_mem_GB": torch.cuda.max_memory_allocated() / 2**30,
               "data": "synthetic, random-init weights"}
        print(json.dumps(out))
        Path("gpurun_out").mkdir(exist_ok=True)
        Path("gpurun_out/train_bench.json").write_text(json.dumps(out, indent=1))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
