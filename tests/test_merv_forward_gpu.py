"""GPU: the product's projector / fusion / splice / training batch assembly against tests/golden/merv_forward.npz -- what the
reference's own MERV.forward body (merv/models/vidlms/merv.py:503-734, executed by tools/make_goldens.py gen_merv_forward)
handed to llm_backbone(inputs_embeds=..., attention_mask=..., labels=...). Tolerances: masks / labels / text rows bit-exact
(the text rows are bf16(embedding), a copy); visual span rel-L2 <= 1e-2 against the fp32 reference (bf16 kernels)."""
import json
import math
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"
FUS_KEYS = ("Q", "attention.q_proj_weight", "attention.k_proj_weight", "attention.in_proj_bias")


def _modules(z, g, dev):
    from merv_amd.projector import AveragePooling3DProjector, CrossAttentionAdapterLearnableQuery
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    projs = []
    for i, C in enumerate(g["C"]):
        p = AveragePooling3DProjector(C, g["llm"], output_frames=g["T"], output_size=g["out_size"], mlp_type="linear")
        p.load_state_dict({"projector.projector.weight": t(f"proj{i}_w"), "projector.projector.bias": t(f"proj{i}_b")})
        projs.append(p.to(dev))
    fus = CrossAttentionAdapterLearnableQuery(embed_dim=g["embed_dim"], llm_dim=g["llm"], token_length=g["T"] * g["out_size"] ** 2, averagetoken=True)
    sd = fus.state_dict()
    for k in FUS_KEYS:
        sd[k] = t(f"fus_{k}")
    fus.load_state_dict(sd)
    return projs, fus.to(dev)


@pytest.mark.parametrize("tag", ["full", "mixed", "nobos", "infer"])
def test_product_tail_matches_what_the_reference_forward_hands_to_the_llm(dev, tag):
    from merv_amd.projector import splice
    from merv_amd.train import FusionFunction, ProjectorFunction, assemble_training_batch, fold_query
    z = np.load(G / "merv_forward.npz")
    meta = json.loads((G / "merv_forward.json").read_text())
    g, m = meta["geometry"], meta[tag]
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    projs, fus = _modules(z, g, dev)
    mm = torch.tensor(m["multimodal_indices"]) if m["multimodal_indices"] is not None else torch.arange(m["B"])
    feats = [t(f"{tag}_feat{i}").float()[mm].reshape(-1, g["T"], s, C).to(dev) for i, (s, C) in enumerate(zip(g["S"], g["C"]))]  # merv.py:571-585
    want_fused, want_w = t(f"{tag}_fused"), t(f"{tag}_fusion_weights")
    # inference modules (MERVVisual.encode's tail) and the autograd functions of the training step: same kernels, both checked
    with torch.no_grad():
        fused_inf, w_inf = fus([p(f) for p, f in zip(projs, feats)])
    projected = [ProjectorFunction.apply(f.to(torch.bfloat16), p.projector.projector.weight, p.projector.projector.bias, p.output_size)
                 for p, f in zip(projs, feats)]
    fused_tr, w_tr = FusionFunction.apply(fold_query(fus), *projected)
    for fused, w in ((fused_inf, w_inf), (fused_tr, w_tr)):
        assert rel_l2(fused, want_fused) < 1e-2
        assert float((w.float().cpu() - want_w).abs().max()) < 5e-3
    emb = t("embed_table")[t(f"{tag}_input_ids")].to(dev)
    bos, S, Tv = m["bos_token_length"], m["S"], g["T"] * g["out_size"] ** 2
    want_e = t(f"{tag}_inputs_embeds")
    if m["with_masks"]:
        e, a, l = assemble_training_batch(emb, fused_tr.detach(), t(f"{tag}_attention_mask").to(dev), t(f"{tag}_labels").to(dev), mm.to(dev), bos)
        assert torch.equal(a.cpu(), t(f"{tag}_out_attention_mask")) and torch.equal(l.cpu(), t(f"{tag}_out_labels"))
    else:
        e = splice(emb[mm.to(dev)], fused_inf, bos)  # merv_splice_forward
    e = e.float().cpu()
    assert e.shape == want_e.shape
    n_mm = len(mm)
    text = torch.cat([want_e[:n_mm, :bos], want_e[:n_mm, bos + Tv:]], 1)
    got_text = torch.cat([e[:n_mm, :bos], e[:n_mm, bos + Tv:]], 1)
    assert torch.equal(got_text, text.to(torch.bfloat16).float())  # copies of bf16(embedding rows), bit for bit
    assert rel_l2(e[:n_mm, bos:bos + Tv], want_e[:n_mm, bos:bos + Tv]) < 1e-2
    if n_mm < m["B"]:  # unimodal rows: text then a zero span, below the multimodal rows (merv.py:676-719)
        assert torch.equal(e[n_mm:, :S], want_e[n_mm:, :S].to(torch.bfloat16).float()) and float(e[n_mm:, S:].abs().max()) == 0.0
