"""
Data-parallel training step for the trainable tail of the visual path (BASELINE.json
configs[3]: "merv-full data-parallel training step (projector+LLM unfrozen)").

What the reference does (merv/training/strategies/base_strategy.py run_training; fsdp.py:208-322; merv.py:305-384,
:562-734): encoders frozen under no_grad, projectors + fusion (+ the LLM in the finetune stages) trainable, bf16
autocast, loss from the HF causal LM on the spliced embeddings with the visual span and the prompt masked by
IGNORE_INDEX, AdamW with decay / no-decay groups, linear-warmup + cosine decay, global grad-norm clipping, FSDP
sharding because an 80 GB device cannot hold a 7B model's fp32 master weights + Adam moments + gradients.

What this build does instead:
  * encoders run on the HIP path, forward-only, concurrently on their streams;
  * projector and fusion are `torch.autograd.Function`s whose forward AND backward are the library's HIP kernels
    (merv_projector_forward/backward, merv_fusion_forward/backward_*; include/merv_hip.h) -- autograd only carries
    the graph; the B x E softmax-backward scalars and the 3072-wide query fold stay in torch;
  * the LLM stays PyTorch-ROCm (north_star), fp32 master weights under bf16 autocast;
  * one process per GPU, plain data parallelism: 288 GB of HBM3E holds the whole 7B training state
    (28 GB weights + 28 GB gradients + 56 GB Adam moments), so no parameter sharding and no all-gathers in the
    step; gradients live in ONE flat buffer per dtype and are averaged by a few large RCCL all-reduces issued
    bucket by bucket (xGMI rings are per-link bound: few large messages, not many small ones).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr
from .projector import AveragePooling3DProjector, CrossAttentionAdapterLearnableQuery
from .vidlm import IGNORE_INDEX, MERV


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


# ---------------------------------------------------------------------------------------------------------------
# autograd wrappers: HIP forward + HIP backward
# ---------------------------------------------------------------------------------------------------------------
class ProjectorFunction(torch.autograd.Function):
    """AveragePooling3DProjector.forward (nn_utils.py:320-330) with its gradient w.r.t. the Linear's weight and bias.
    The pooled encoder tokens are kept for backward; the encoder features get no gradient (frozen, merv.py:562)."""

    @staticmethod
    def forward(ctx, feats: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, out_size: int) -> torch.Tensor:
        lib = _lib.load()
        B, Fr, N, Cc = feats.shape
        H = int(math.isqrt(N))
        x = feats.detach().to(torch.bfloat16).contiguous()
        w = weight.detach().to(torch.bfloat16).contiguous()
        b = bias.detach().to(torch.float32).contiguous()
        llm = w.shape[0]
        M = B * Fr * out_size * out_size
        pooled = torch.empty(M, Cc, dtype=torch.bfloat16, device=x.device)
        out = torch.empty(B, Fr * out_size * out_size, llm, dtype=torch.bfloat16, device=x.device)
        check(lib.merv_projector_forward(ptr(x), B, Fr, H, Cc, out_size, ptr(w), ptr(b), llm, ptr(pooled), ptr(out),
                                         _stream(x.device)), "merv_projector_forward")
        ctx.save_for_backward(pooled)
        ctx.meta = (M, Cc, llm, weight.dtype, bias.dtype)
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        lib = _lib.load()
        (pooled,) = ctx.saved_tensors
        M, Cc, llm, wdt, bdt = ctx.meta
        g = grad_out.to(torch.bfloat16).contiguous().view(M, llm)
        nbytes = lib.merv_projector_backward_workspace_bytes(M, Cc, llm)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
        gw = torch.empty(llm, Cc, dtype=torch.bfloat16, device=g.device)
        gb = torch.empty(llm, dtype=torch.float32, device=g.device)
        check(lib.merv_projector_backward(ptr(g), ptr(pooled), M, Cc, llm, ptr(ws), nbytes, ptr(gw), ptr(gb),
                                          _stream(g.device)), "merv_projector_backward")
        return None, gw.to(wdt), gb.to(bdt), None


class FusionFunction(torch.autograd.Function):
    """CrossAttentionAdapterLearnableQuery.forward, averagetoken=True (nn_utils.py:487-521), on the folded query `u`
    (differentiable w.r.t. u and every V_e). Returns (fused, weights); weights carry no gradient (the reference only
    logs them, merv.py:607-612)."""

    @staticmethod
    def forward(ctx, u: torch.Tensor, *V: torch.Tensor):
        lib = _lib.load()
        E = len(V)
        B, T, Cc = V[0].shape
        dev = V[0].device
        Vc = [v.detach().to(torch.bfloat16).contiguous() for v in V]
        u32 = u.detach().to(torch.float32).contiguous()
        partial = torch.empty(lib.merv_fusion_workspace_floats(B, E, T), dtype=torch.float32, device=dev)
        w = torch.empty(B, E, dtype=torch.float32, device=dev)
        out = torch.empty(B, T, Cc, dtype=torch.bfloat16, device=dev)
        arr = (C.c_void_p * E)(*[ptr(v) for v in Vc])
        check(lib.merv_fusion_forward(arr, E, B, T, Cc, ptr(u32), ptr(partial), ptr(w), ptr(out), _stream(dev)),
              "merv_fusion_forward")
        ctx.save_for_backward(u32, w, *Vc)
        ctx.mark_non_differentiable(w)
        ctx.u_dtype = u.dtype
        return out, w

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor, _grad_w):
        lib = _lib.load()
        u32, w, *Vc = ctx.saved_tensors
        E = len(Vc)
        B, T, Cc = Vc[0].shape
        dev = grad_out.device
        g = grad_out.to(torch.bfloat16).contiguous()
        ws = torch.empty(lib.merv_fusion_backward_workspace_floats(B, E, T, Cc), dtype=torch.float32, device=dev)
        dw = torch.empty(B, E, dtype=torch.float32, device=dev)
        vbar = torch.empty(B, E, Cc, dtype=torch.float32, device=dev)
        arr = (C.c_void_p * E)(*[ptr(v) for v in Vc])
        check(lib.merv_fusion_backward_reduce(arr, E, B, T, Cc, ptr(g), ptr(ws), ptr(dw), ptr(vbar), _stream(dev)),
              "merv_fusion_backward_reduce")
        ds = (w * (dw - (w * dw).sum(-1, keepdim=True))).contiguous()  # softmax backward over the E encoders
        du = torch.einsum("be,bec->c", ds, vbar)
        dV = [torch.empty_like(v) for v in Vc]
        out_arr = (C.c_void_p * E)(*[ptr(v) for v in dV])
        check(lib.merv_fusion_backward_mix(ptr(g), ptr(w), ptr(ds), ptr(u32), E, B, T, Cc, out_arr, _stream(dev)),
              "merv_fusion_backward_mix")
        return (du.to(ctx.u_dtype), *dV)


class SpliceFunction(torch.autograd.Function):
    """merv.py:633-640 as the HIP copy kernel; backward hands each source its slice of the gradient."""

    @staticmethod
    def forward(ctx, emb: torch.Tensor, fused: torch.Tensor, bos: int) -> torch.Tensor:
        from .projector import splice
        ctx.meta = (bos, fused.shape[1], emb.dtype, fused.dtype)
        return splice(emb.detach(), fused.detach(), bos)

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        bos, T, edt, fdt = ctx.meta
        g_emb = torch.cat([g[:, :bos], g[:, bos + T:]], dim=1).to(edt)
        return g_emb, g[:, bos:bos + T].to(fdt), None


def fold_query(fusion: CrossAttentionAdapterLearnableQuery) -> torch.Tensor:
    """u = Wk^T (Wq Q + bq) / sqrt(embed_dim) in fp32 on the parameters' device, differentiable w.r.t. Q, Wq, bq, Wk.
    (bk, Wv, bv and out_proj get no gradient: they only feed the MHA output the reference discards, nn_utils.py:512.)"""
    a = fusion.attention
    Ed = fusion.Q.shape[1]
    q = a.q_proj_weight.float() @ fusion.Q.float()[0] + a.in_proj_bias.float()[:Ed]
    return (a.k_proj_weight.float().t() @ q) / math.sqrt(Ed)


# ---------------------------------------------------------------------------------------------------------------
# training forward (merv.py:562-734)
# ---------------------------------------------------------------------------------------------------------------
def encode_trainable(vidlm: MERV, video_values: Sequence[torch.Tensor]):
    """Encoders forward-only on their streams, then projector / fusion with gradients. Returns (fused, weights|None)."""
    # a4-a8 through the one orchestration (MervVisualPath: four encoder streams, persistent buffers); the tokens are
    # cloned because autograd keeps them until backward while the path's buffers are reused by the next forward
    with torch.no_grad():
        toks = vidlm.visual_path(video_values[0].device).encode_tokens(video_values)
        feats = [t.clone().reshape(-1, vb.temporal_resolution, vb.spatial_resolution, t.shape[-1])  # merv.py:576-585
                 for t, vb in zip(toks, vidlm.video_backbones)]
    projected = []
    for f, proj in zip(feats, vidlm.projectors):
        lin = proj.projector.projector
        projected.append(ProjectorFunction.apply(f, lin.weight, lin.bias, proj.output_size))
    if vidlm.feature_fusion is None:
        if len(projected) != 1:
            raise TypeError("argument of type 'NoneType' is not iterable")  # reference behaviour, merv.py:607
        return projected[0], None
    T = vidlm.feature_fusion.token_length
    projected = [(p.repeat(1, T, 1) if p.shape[1] == 1 else p) for p in projected]
    return FusionFunction.apply(fold_query(vidlm.feature_fusion), *projected)


def assemble_training_batch(input_embeddings: torch.Tensor, fused: torch.Tensor, attention_mask: torch.Tensor,
                            labels: torch.Tensor, multimodal_indices: torch.Tensor, bos_token_length: int = 1):
    """merv.py:612-719: multimodal rows get the visual span after BOS (mask True, labels IGNORE_INDEX); unimodal rows are
    padded at the END by the same length (mask False, labels IGNORE_INDEX) and stacked below the multimodal ones."""
    mm = multimodal_indices
    b = bos_token_length
    Tv = fused.shape[1]
    emb_mm = SpliceFunction.apply(input_embeddings[mm], fused, b)
    vis_mask = torch.full((len(mm), Tv), True, dtype=attention_mask.dtype, device=attention_mask.device)
    am_mm = torch.cat([attention_mask[mm, :b], vis_mask, attention_mask[mm, b:]], dim=1)
    vis_lab = torch.full((len(mm), Tv), IGNORE_INDEX, dtype=labels.dtype, device=labels.device)
    lab_mm = torch.cat([labels[mm, :b], vis_lab, labels[mm, b:]], dim=1)
    mm_set = set(mm.tolist())
    uni = torch.tensor([i for i in range(input_embeddings.shape[0]) if i not in mm_set], dtype=torch.long, device=mm.device)
    if len(uni) == 0:
        return emb_mm, am_mm, lab_mm
    n = len(uni)
    emb_u = torch.cat([input_embeddings[uni].to(emb_mm.dtype),
                       torch.zeros(n, Tv, input_embeddings.shape[2], dtype=emb_mm.dtype, device=emb_mm.device)], dim=1)
    am_u = torch.cat([attention_mask[uni], torch.full((n, Tv), False, dtype=attention_mask.dtype, device=attention_mask.device)], dim=1)
    lab_u = torch.cat([labels[uni], torch.full((n, Tv), IGNORE_INDEX, dtype=labels.dtype, device=labels.device)], dim=1)
    return torch.vstack([emb_mm, emb_u]), torch.vstack([am_mm, am_u]), torch.vstack([lab_mm, lab_u])


def training_forward(vidlm: MERV, input_ids: torch.Tensor, attention_mask: torch.Tensor, video_values: Sequence[torch.Tensor],
                     labels: torch.Tensor, multimodal_indices: Optional[torch.Tensor] = None):
    """MERV.forward in training mode (merv.py:503-734): returns (loss, logits, fusion_weights)."""
    llm = vidlm.llm_backbone
    if multimodal_indices is None:  # merv.py:543-545
        multimodal_indices = torch.arange(len(input_ids), dtype=torch.long, device=input_ids.device)
    if len(multimodal_indices) == 0:  # merv.py:548-561: plain language-model forward
        with torch.autocast("cuda", dtype=torch.bfloat16):  # same mixed-precision context as the multimodal branch
            out = llm.llm(input_ids=input_ids, attention_mask=attention_mask, labels=labels)
        return out.loss, out.logits, None
    from .vidlm import bos_token_length
    bos = bos_token_length(llm, getattr(vidlm, "tokenizer", None))  # merv.py:520-521: the tokenizer decides
    fused, w = encode_trainable(vidlm, [v[multimodal_indices] for v in video_values])  # merv.py:563-566
    emb = llm.llm.get_input_embeddings()(input_ids)
    emb_all, am_all, lab_all = assemble_training_batch(emb, fused, attention_mask, labels, multimodal_indices, bos)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = llm.llm(inputs_embeds=emb_all, attention_mask=am_all, labels=lab_all)
    return out.loss, out.logits, w


# ---------------------------------------------------------------------------------------------------------------
# stages, optimizer groups, schedule (merv.py:305-384, fsdp.py:262-293)
# ---------------------------------------------------------------------------------------------------------------
STAGE_MODULES = {
    "align": ["projectors", "feature_fusion"],
    "full-align": ["projectors", "llm_backbone", "feature_fusion"],
    "finetune": ["projectors", "llm_backbone", "feature_fusion"],
    "second_finetune": ["projectors", "llm_backbone", "feature_fusion"],
}


def freeze_backbones(vidlm: MERV, stage: str) -> List[str]:
    """merv.py:305-384: which modules train in which stage; the video encoders never do on this path."""
    if stage not in STAGE_MODULES:
        raise ValueError(f"Stage `{stage}` is not supported for MERV! Try < align | finetune >")
    keys = [k for k in STAGE_MODULES[stage] if not (k == "feature_fusion" and vidlm.feature_fusion is None)]
    vidlm.projectors.requires_grad_(True)
    if vidlm.feature_fusion is not None:
        vidlm.feature_fusion.requires_grad_(True)
    vidlm.llm_backbone.llm.requires_grad_("llm_backbone" in keys)
    vidlm.trainable_module_keys = keys
    return keys


def trainable_named_parameters(vidlm: MERV) -> List[Tuple[str, nn.Parameter]]:
    named = [("projectors." + n, p) for n, p in vidlm.projectors.named_parameters()]
    if vidlm.feature_fusion is not None:
        named += [("feature_fusion." + n, p) for n, p in vidlm.feature_fusion.named_parameters()]
    named += [("llm_backbone.llm." + n, p) for n, p in vidlm.llm_backbone.llm.named_parameters()]
    return [(n, p) for n, p in named if p.requires_grad]


def build_optimizer(named_params: Iterable[Tuple[str, nn.Parameter]], learning_rate: float, weight_decay: float):
    """fsdp.py:276-290: AdamW; parameters with ndim <= 1 or a name ending in `.bias` are not decayed. On a GPU the
    update runs as torch's fused multi-tensor kernel (one pass over the 7B-parameter state instead of a launch per
    tensor); the arithmetic is AdamW's either way."""
    decay, no_decay = [], []
    for name, p in named_params:
        (no_decay if (p.ndim <= 1 or name.endswith(".bias")) else decay).append(p)
    groups = [{"params": decay, "weight_decay": weight_decay}, {"params": no_decay, "weight_decay": 0.0}]
    on_gpu = all(p.is_cuda for g in groups for p in g["params"]) and any(len(g["params"]) for g in groups)
    return torch.optim.AdamW(groups, lr=learning_rate, fused=True) if on_gpu else torch.optim.AdamW(groups, lr=learning_rate)


def cosine_with_warmup(step: int, num_warmup_steps: int, num_training_steps: int) -> float:
    """transformers.get_cosine_schedule_with_warmup's multiplier (the scheduler fsdp.py:291 builds), half a cosine."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))


# ---------------------------------------------------------------------------------------------------------------
# gradient exchange: one flat buffer, few large all-reduces
# ---------------------------------------------------------------------------------------------------------------
class FlatGradSync:
    """Gradients of all trainable parameters are views into one flat fp32 buffer (set once, before the first
    backward), so the data-parallel average is `ceil(bytes / bucket)` all-reduces over contiguous memory -- no
    per-parameter launches, no packing copies. Bucket size default 256 MiB: a ring all-reduce over xGMI moves
    2(N-1)/N of the bucket over each ≈153 GB/s link, ≈3 ms per bucket at N=8, launch cost amortised to noise."""

    def __init__(self, params: Sequence[nn.Parameter], process_group=None, bucket_bytes: int = 256 << 20) -> None:
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.bucket_elems = max(1, bucket_bytes // 4)

    def zero(self) -> None:
        self.flat.zero_()

    def check_views(self) -> None:
        base = self.flat.untyped_storage().data_ptr()
        for p in self.params:
            if p.grad is None or p.grad.untyped_storage().data_ptr() != base:
                raise RuntimeError("a gradient was re-allocated outside the flat buffer (use FlatGradSync.zero(), not "
                                   "optimizer.zero_grad(set_to_none=True))")

    def all_reduce_mean(self) -> None:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.group)
        if world == 1:
            return
        self.check_views()
        works = []
        for s in range(0, self.flat.numel(), self.bucket_elems):
            works.append(dist.all_reduce(self.flat[s:s + self.bucket_elems], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        self.flat.div_(world)

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """Global L2 clipping over the (already averaged) flat buffer: identical on every rank, no extra collective."""
        total = self.flat.norm(2)
        scale = (max_norm / (total + 1e-6)).clamp(max=1.0)
        self.flat.mul_(scale)
        return total


class TrainStep:
    """One optimisation step of run_training (base_strategy.py) for a per-rank micro-batch: forward, backward, RCCL
    gradient average, clip, AdamW, schedule. Gradient accumulation = call `accumulate()` k-1 times, then `step()`."""

    def __init__(self, vidlm: MERV, stage: str = "finetune", learning_rate: float = 2e-5, weight_decay: float = 0.1,
                 max_grad_norm: float = 1.0, warmup_ratio: float = 0.03, max_steps: int = 1000, grad_accumulation_steps: int = 1,
                 process_group=None, bucket_bytes: int = 256 << 20, enable_gradient_checkpointing: bool = True) -> None:
        self.vidlm = vidlm
        freeze_backbones(vidlm, stage)
        if enable_gradient_checkpointing and "llm_backbone" in vidlm.trainable_module_keys:
            # conf/models.py:79 default; the reference wraps each decoder layer (fsdp.py:243-255), HF's own switch does
            # the same without FSDP in the way
            vidlm.llm_backbone.llm.config.use_cache = False
            vidlm.llm_backbone.llm.gradient_checkpointing_enable(gradient_checkpointing_kwargs={"use_reentrant": False})
            vidlm.llm_backbone.llm.train()
        named = trainable_named_parameters(vidlm)
        if any(p.dtype != torch.float32 for _, p in named):
            raise ValueError("trainable parameters must be fp32 master weights (build the LLM with dtype=torch.float32)")
        self.optimizer = build_optimizer(named, learning_rate, weight_decay)
        self.sync = FlatGradSync([p for _, p in named], process_group, bucket_bytes)
        self.base_lr, self.max_grad_norm = learning_rate, max_grad_norm
        self.num_training_steps = max_steps
        self.num_warmup_steps = int(max_steps * warmup_ratio)  # fsdp.py:272
        self.grad_accumulation_steps = grad_accumulation_steps
        self.global_step = 0
        self._set_lr()

    def _set_lr(self) -> None:
        lr = self.base_lr * cosine_with_warmup(self.global_step, self.num_warmup_steps, self.num_training_steps)
        for g in self.optimizer.param_groups:
            g["lr"] = lr

    def accumulate(self, batch: Dict) -> torch.Tensor:
        loss, _, _ = training_forward(self.vidlm, batch["input_ids"], batch["attention_mask"], batch["video_values"],
                                      batch["labels"], batch.get("multimodal_indices"))
        (loss / self.grad_accumulation_steps).backward()
        return loss.detach()

    def step(self, batch: Dict) -> Dict[str, float]:
        loss = self.accumulate(batch)
        self.sync.all_reduce_mean()
        gnorm = self.sync.clip_grad_norm_(self.max_grad_norm)
        self.optimizer.step()
        self.sync.zero()
        for proj in self.vidlm.projectors:  # inference-side bf16 copies of the updated parameters are stale now
            proj._dev = None
        if self.vidlm.feature_fusion is not None:
            self.vidlm.feature_fusion._u = None
        self.global_step += 1
        self._set_lr()
        return {"loss": float(loss), "grad_norm": float(gnorm), "lr": self.optimizer.param_groups[0]["lr"]}
