"""GPU: backward kernels and the training step against torch autograd through the oracle --
what the reference's loss.backward() differentiates. Tolerances: gradients are bf16 products with fp32 accumulation over
hundreds to thousands of rows; relative L2 error bounds are written at each comparison."""
import ctypes as C

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _s(dev):
    return torch.cuda.current_stream(dev).cuda_stream


@pytest.mark.parametrize("R,Cc", [(64, 64), (1, 2), (70, 130), (513, 258), (2048, 1024)])
def test_transpose_bit_exact(dev, R, Cc):
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(R * 7 + Cc)
    x = torch.randn(R, Cc, generator=g, device=dev).to(torch.bfloat16)
    Rpad = (R + 63) // 64 * 64
    out = torch.full((Cc, Rpad), 7.0, dtype=torch.bfloat16, device=dev)
    check(lib.merv_transpose_bf16(ptr(x), R, Cc, Cc, ptr(out), Rpad, Rpad, _s(dev)), "transpose")
    assert torch.equal(out[:, :R], x.t())
    assert not out[:, R:].any()  # padding columns are zero-filled
    with pytest.raises(ValueError):
        check(lib.merv_transpose_bf16(ptr(x), R, Cc, Cc, ptr(out), Rpad, Rpad + 1, _s(dev)), "transpose")


@pytest.mark.parametrize("B,Fr,S,Cc,llm", [(2, 4, 16, 128, 256), (1, 2, 14, 256, 384), (3, 16, 16, 128, 512)])
def test_projector_backward_vs_oracle_autograd(dev, B, Fr, S, Cc, llm):
    from oracle import merv_oracle as O
    from merv_amd.train import ProjectorFunction
    g = torch.Generator().manual_seed(B * 100 + S)
    feats = torch.randn(B, Fr, S * S, Cc, generator=g).to(torch.bfloat16)
    w = (torch.randn(llm, Cc, generator=g) * Cc**-0.5).requires_grad_()
    b = torch.randn(llm, generator=g).requires_grad_()
    gout = torch.randn(B, Fr * 64, llm, generator=g).to(torch.bfloat16)
    # oracle: autograd through the restated projector (bf16-rounded weights, as the HIP forward consumes them)
    ref = O.projector_forward(feats.float().reshape(B, Fr * S * S, Cc), Fr, S, 8, w, b)
    ref.backward(gout.float())
    wd = w.detach().clone().to(dev).requires_grad_()
    bd = b.detach().clone().to(dev).requires_grad_()
    out = ProjectorFunction.apply(feats.to(dev), wd, bd, 8)
    assert rel_l2(out.detach(), ref.detach()) < 1e-2
    out.backward(gout.to(dev))
    assert wd.grad.dtype == torch.float32 and wd.grad.shape == (llm, Cc)
    assert rel_l2(wd.grad, w.grad) < 1e-2  # bf16 output of an fp32-accumulated K = B*Fr*64 contraction
    assert rel_l2(bd.grad, b.grad) < 2e-3  # fp32 column sums of bf16 values


@pytest.mark.parametrize("B,E,T,Cc,Ed", [(2, 4, 64, 256, 96), (1, 3, 50, 128, 64), (3, 4, 1024, 512, 128), (2, 1, 16, 64, 32)])
def test_fusion_backward_vs_oracle_autograd(dev, B, E, T, Cc, Ed):
    from oracle import merv_oracle as O
    from merv_amd.projector import CrossAttentionAdapterLearnableQuery
    from merv_amd.train import FusionFunction, fold_query
    torch.manual_seed(B * 10 + E)
    m = CrossAttentionAdapterLearnableQuery(embed_dim=Ed, llm_dim=Cc, token_length=T, averagetoken=True, num_encoder=E)
    with torch.no_grad():
        m.Q.mul_(40)  # far from uniform softmax weights
        m.attention.in_proj_bias.normal_(0, 0.5)
    g = torch.Generator().manual_seed(5)
    V = [torch.randn(B, T, Cc, generator=g).to(torch.bfloat16) for _ in range(E)]
    gout = torch.randn(B, T, Cc, generator=g).to(torch.bfloat16)
    # oracle autograd
    Fw = {k: v.detach().clone().requires_grad_() for k, v in m.state_dict().items()}
    Vr = [v.float().requires_grad_() for v in V]
    ref, wref = O.fusion_forward(Vr, Fw)
    ref.backward(gout.float())
    # HIP forward + backward
    md = m.to(dev)
    Vd = [v.to(dev).requires_grad_() for v in V]
    out, w = FusionFunction.apply(fold_query(md), *Vd)
    assert (w.cpu() - wref.detach()).abs().max() < 5e-3 and not w.requires_grad
    if E > 1:
        assert float(wref.detach().max()) < 0.999 and float(wref.detach().std()) > 0.02, "degenerate softmax: test would be vacuous"
    assert rel_l2(out.detach(), ref.detach()) < 1e-2
    out.backward(gout.to(dev))
    for e in range(E):
        assert rel_l2(Vd[e].grad, Vr[e].grad) < 1e-2, e  # bf16 rounding of w_e g + ds_e u / T
    if E > 1:
        a = md.attention
        Edim = m.Q.shape[1]
        assert rel_l2(md.Q.grad, Fw["Q"].grad) < 2e-2
        assert rel_l2(a.q_proj_weight.grad, Fw["attention.q_proj_weight"].grad) < 2e-2
        assert rel_l2(a.k_proj_weight.grad, Fw["attention.k_proj_weight"].grad) < 2e-2
        assert rel_l2(a.in_proj_bias.grad[:Edim], Fw["attention.in_proj_bias"].grad[:Edim]) < 2e-2
        # parameters that only feed the discarded MHA output get no gradient on either side (nn_utils.py:512)
        assert a.v_proj_weight.grad is None and Fw["attention.v_proj_weight"].grad is None


TINY_LLM = dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=4, max_position_embeddings=4096, rms_norm_eps=1e-5, bos_token_id=1, eos_token_id=2,
                pad_token_id=0)


def _tiny_merv(dev, seed=0):
    from merv_amd.backbones import VIDEO_BACKBONES
    from merv_amd.llm import LlamaBackbone
    from merv_amd.vidlm import MERV
    ids = ["dinov2-video-all-tokens", "siglip-vit-b16-224px-all-no-cls"]
    bbs = [VIDEO_BACKBONES[i]["cls"](i, "resize-naive", num_frames=4, weights="random", device=dev, layers=1,
                                     **VIDEO_BACKBONES[i]["kwargs"]) for i in ids]
    llm = LlamaBackbone(TINY_LLM, device=dev, dtype=torch.float32, seed=seed)
    m = MERV(bbs, llm, visual_feature_length=256).to(dev)
    with torch.no_grad():
        m.feature_fusion.Q.mul_(30)
    return m, bbs


def _batch(bbs, dev, n=3, uni=(1,), seed=0):
    g = torch.Generator().manual_seed(seed)
    S = 9
    input_ids = torch.randint(3, 300, (n, S), generator=g)
    input_ids[:, 0] = 1
    input_ids[n - 1, 7:] = 0  # right padding
    attention_mask = input_ids.ne(0)
    labels = input_ids.clone()
    labels[:, :4] = -100
    labels[~attention_mask] = -100
    video_values = [torch.randn(n, *b.default_video_resolution, generator=g) for b in bbs]
    mm = torch.tensor([i for i in range(n) if i not in uni], dtype=torch.long)
    to = lambda t: t.to(dev)
    return dict(input_ids=to(input_ids), attention_mask=to(attention_mask), labels=to(labels),
                video_values=[to(v) for v in video_values], multimodal_indices=to(mm))


def test_training_forward_loss_and_grads_vs_oracle(dev):
    """Mixed multimodal / language-only batch through encoders (HIP) -> projector -> fusion -> assembly -> LLM loss;
    the oracle runs the same parameters in fp32 on the CPU and differentiates with autograd."""
    from oracle import merv_oracle as O
    from merv_amd.backbones import random_weights
    from merv_amd.train import freeze_backbones, training_forward
    m, bbs = _tiny_merv(dev)
    freeze_backbones(m, "finetune")
    batch = _batch(bbs, dev)
    loss, logits, w = training_forward(m, batch["input_ids"], batch["attention_mask"], batch["video_values"], batch["labels"],
                                       batch["multimodal_indices"])
    loss.backward()
    assert logits.shape[:2] == (3, 9 + 256)
    # ---- oracle ----
    mm = batch["multimodal_indices"].cpu()
    projected, pw = [], []
    for b, vv, pr in zip(bbs, batch["video_values"], m.projectors):
        cfg = O.EncoderCfg(**{k: getattr(b.spec, k) for k in O.EncoderCfg.__dataclass_fields__})
        tok = O.encoder_forward(vv.cpu()[mm], cfg, random_weights(b.spec, seed=b.spec.dim + b.spec.frames))
        lin = pr.projector.projector
        wq, bq = lin.weight.detach().cpu().clone().requires_grad_(), lin.bias.detach().cpu().clone().requires_grad_()
        pw.append((wq, bq))
        projected.append(O.projector_forward(tok, 4, b.spec.hp, 8, wq, bq))
    Fw = {k: v.detach().cpu().clone().requires_grad_() for k, v in m.feature_fusion.state_dict().items()}
    fused, wref = O.fusion_forward(projected, Fw)
    import copy
    llm_cpu = copy.deepcopy(m.llm_backbone.llm).cpu().float()
    emb = llm_cpu.get_input_embeddings()(batch["input_ids"].cpu())
    emb_all, am_all, lab_all = O.assemble_training_batch(emb, fused, batch["attention_mask"].cpu(), batch["labels"].cpu(), mm, 1)
    ref_logits = llm_cpu(inputs_embeds=emb_all, attention_mask=am_all).logits
    ref_loss = O.causal_lm_loss(ref_logits, lab_all)
    ref_loss.backward()
    assert (w.cpu() - wref.detach()).abs().max() < 1e-2
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 2e-2 * max(1.0, abs(float(ref_loss.detach())))  # bf16 autocast LLM vs fp32
    cos = torch.nn.functional.cosine_similarity
    for pr, (wq, bq) in zip(m.projectors, pw):
        lin = pr.projector.projector
        assert cos(lin.weight.grad.flatten().cpu(), wq.grad.flatten(), dim=0) > 0.98
        assert cos(lin.bias.grad.flatten().cpu(), bq.grad.flatten(), dim=0) > 0.98
    assert cos(m.feature_fusion.Q.grad.flatten().cpu(), Fw["Q"].grad.flatten(), dim=0) > 0.95
    eg = m.llm_backbone.llm.get_input_embeddings().weight.grad
    assert cos(eg.flatten().cpu(), llm_cpu.get_input_embeddings().weight.grad.flatten(), dim=0) > 0.98
    # encoders stay frozen: nothing in them is a parameter of the training graph
    assert all(not p.requires_grad for b in bbs for p in b.parameters())


def test_train_step_descends_and_keeps_flat_grads(dev):
    from merv_amd.train import TrainStep, cosine_with_warmup
    m, bbs = _tiny_merv(dev, seed=1)
    ts = TrainStep(m, stage="finetune", learning_rate=2e-3, weight_decay=0.1, max_grad_norm=1.0, warmup_ratio=0.25, max_steps=8)
    assert ts.num_warmup_steps == 2 and ts.optimizer.param_groups[0]["lr"] == 0.0  # first step runs at lr 0 (LambdaLR at step 0)
    assert ts.optimizer.param_groups[1]["weight_decay"] == 0.0 and ts.optimizer.param_groups[0]["weight_decay"] == 0.1
    batch = _batch(bbs, dev, n=2, uni=(), seed=3)
    losses = []
    for i in range(6):
        before = m.projectors[0].projector.projector.weight.detach().clone()
        info = ts.step(batch)
        losses.append(info["loss"])
        assert abs(info["lr"] - 2e-3 * cosine_with_warmup(i + 1, 2, 8)) < 1e-12
        if i == 0:
            assert torch.equal(before, m.projectors[0].projector.projector.weight)  # lr was 0
        ts.sync.check_views()
        assert float(ts.sync.flat.abs().max()) == 0.0  # zeroed for the next step
    assert losses[-1] < losses[0] - 0.05, losses
    # align stage: the LLM is frozen
    ts2 = TrainStep(m, stage="align", learning_rate=1e-3, max_steps=4)
    assert all(not p.requires_grad for p in m.llm_backbone.llm.parameters())
    assert m.trainable_module_keys == ["projectors", "feature_fusion"]
    ts2.step(batch)
    with pytest.raises(ValueError):
        TrainStep(m, stage="full-finetune")
