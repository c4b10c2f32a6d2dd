"""GPU: the HIP path (through the C ABI) against the committed golden vectors that came out of the REFERENCE's own
modules (tools/make_goldens.py): LanguageBind vision tower, HF VivitModel, nn_utils projector + fusion, and the HF
DINOv2 / SigLIP cross-checks. Tolerance: bf16 operands vs fp32 reference outputs, rel-L2 <= 2e-2 (2 blocks)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def _spec(name, dim, heads, mlp, layers, patch, tub, img, frames, layout, prefix, joint, pre_ln, final_ln, ls, tf, act, eps):
    from merv_amd.encoder import EncoderSpec
    return EncoderSpec(name, dim, heads, mlp, layers, patch, tub, img, frames, layout, prefix, joint, pre_ln, final_ln, ls, tf,
                       act, eps)


@pytest.mark.parametrize("tag,act", [("gelu", "gelu_erf"), ("quick", "quick_gelu")])
def test_languagebind_reference_tower(dev, tag, act):
    from merv_amd import weights as Wm
    from merv_amd.encoder import HipEncoder
    z = np.load(G / "languagebind.npz")
    sd = {k[len(tag) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}_sd/")}
    spec = _spec("languagebind", 128, 2, 256, 2, 14, 1, 56, 16, "BCFHW", 1, False, True, False, False, 8, act, 1e-5)
    enc = HipEncoder(spec, Wm.from_languagebind_vision(sd, n_layers=2), dev)
    out = enc.forward(torch.from_numpy(z[f"{tag}_pix"]).to(dev))
    ref = torch.from_numpy(z[f"{tag}_hs_m2"])[:, :, 1:].reshape(2, -1, 128)
    assert rel_l2(out, ref) < 2e-2


def test_vivit_reference_model(dev):
    from merv_amd import weights as Wm
    from merv_amd.encoder import HipEncoder
    z = np.load(G / "vivit.npz")
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    spec = _spec("vivit", 128, 2, 256, 2, 16, 2, 64, 8, "BFCHW", 1, True, False, True, False, 0, "gelu_tanh", 1e-6)
    enc = HipEncoder(spec, Wm.from_hf_vivit(sd), dev)
    out = enc.forward(torch.from_numpy(z["pix"]).to(dev))
    assert rel_l2(out, torch.from_numpy(z["last_hidden_state"])[:, 1:]) < 2e-2


def test_dinov2_and_siglip_hf_crosscheck(dev):
    from merv_amd import weights as Wm
    from merv_amd.encoder import HipEncoder
    z = np.load(G / "hf_crosscheck.npz")
    sd = {k[8:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("dino_sd/")}
    spec = _spec("dinov2", 128, 2, 256, 2, 14, 1, 56, 1, "BFCHW", 5, False, False, False, True, 0, "gelu_erf", 1e-6)
    out = HipEncoder(spec, Wm.from_hf_dinov2(sd), dev).forward(torch.from_numpy(z["dino_pix"])[:, None].contiguous().to(dev))
    assert rel_l2(out, torch.from_numpy(z["dino_hs"])[2][:, 5:]) < 2e-2
    sd = {k[10:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("siglip_sd/")}
    sd = {k[len("vision_model."):] if k.startswith("vision_model.") else k: v for k, v in sd.items()}
    spec = _spec("siglip", 128, 2, 256, 2, 16, 1, 64, 1, "BFCHW", 0, False, False, False, False, 0, "gelu_erf", 1e-6)
    out = HipEncoder(spec, Wm.from_hf_siglip(sd), dev).forward(torch.from_numpy(z["siglip_pix"])[:, None].contiguous().to(dev))
    assert rel_l2(out, torch.from_numpy(z["siglip_hs"])[2]) < 2e-2


def test_projector_pools_up_like_the_reference_class(dev):
    """merv_projector_forward with S < out_size (round 5: one-token-per-frame selections through MERV) against the reference class's
    output on the same input and parameters (tests/golden/projector_upsample.npz)."""
    from merv_amd.projector import AveragePooling3DProjector
    z = np.load(G / "projector_upsample.npz")
    for tag, T, C in (("s1", 16, 64), ("s2", 8, 64)):
        proj = AveragePooling3DProjector(C, 128, output_frames=T, output_size=8, mlp_type="linear")
        proj.load_state_dict({"projector.projector.weight": torch.from_numpy(z[f"{tag}_w"]),
                              "projector.projector.bias": torch.from_numpy(z[f"{tag}_b"])})
        out = proj(torch.from_numpy(z[f"{tag}_x"]).to(dev))
        assert rel_l2(out, torch.from_numpy(z[f"{tag}_y"])) < 1e-2, tag


def test_projector_and_fusion_reference_classes(dev):
    from merv_amd.projector import AveragePooling3DProjector, CrossAttentionAdapterLearnableQuery
    z = np.load(G / "projector_fusion.npz")
    x = torch.from_numpy(z["proj_s16_x"])  # [B, T, 256, 64]
    proj = AveragePooling3DProjector(64, 128, output_frames=16, output_size=8, mlp_type="linear")
    proj.load_state_dict({"projector.projector.weight": torch.from_numpy(z["proj_s16_w"]),
                          "projector.projector.bias": torch.from_numpy(z["proj_s16_b"])})  # the reference's own keys
    out = proj(x.to(dev))
    assert rel_l2(out, torch.from_numpy(z["proj_s16_y"])) < 1e-2
    for tag in ("small", "e1", "wide"):
        Q = torch.from_numpy(z[f"fus_{tag}_Q"])
        kw = torch.from_numpy(z[f"fus_{tag}_attention.k_proj_weight"])
        V = [torch.from_numpy(v) for v in z[f"fus_{tag}_V"]]
        fus = CrossAttentionAdapterLearnableQuery(embed_dim=Q.shape[1], llm_dim=kw.shape[1], token_length=V[0].shape[1],
                                                  averagetoken=True)
        sd = fus.state_dict()
        for k in ("Q", "attention.q_proj_weight", "attention.k_proj_weight", "attention.in_proj_bias"):
            sd[k] = torch.from_numpy(z[f"fus_{tag}_{k}"])
        fus.load_state_dict(sd)
        out, w = fus([v.to(dev) for v in V])
        assert (w.cpu() - torch.from_numpy(z[f"fus_{tag}_w"])).abs().max() < 5e-3, tag
        assert rel_l2(out, torch.from_numpy(z[f"fus_{tag}_y"])) < 1e-2, tag
