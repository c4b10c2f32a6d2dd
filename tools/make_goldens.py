#!/usr/bin/env python3
"""
make_goldens.py -- generates tests/golden/* by running the REFERENCE's own code in this container.

Runs only where /root/reference exists (the build container); the GPU box and the test-suite use the committed
fixtures. Nothing of the reference's source is copied: the script imports reference modules *by file path* with
tiny stubs for packages the image lacks (timm, peft) and stores input/output vectors only.

What is pinned, and against what:
  frame_indices.json   numpy.linspace(..., dtype=int) evaluated with the argument expressions of
                       merv/preprocessing/datasets/datasets.py:131-141 (decord itself is absent, so the two
                       np.linspace call sites are evaluated directly), incl. eval_data/dummy_mcq end_frame=595.
  projector_fusion.npz the reference's AveragePooling3DProjector / CrossAttentionAdapterLearnableQuery
                       (merv/util/nn_utils.py, imported by path) -- inputs, state dicts, outputs.
  languagebind.npz     the reference's vendored CLIPVisionTransformer (languagebind/video/modeling_video.py, imported
                       by path; composes the installed transformers' CLIPAttention/CLIPMLP/CLIPVisionEmbeddings),
                       2 layers, D=128, T=16 frames with config.num_frames=8 -> hidden_states[-2].
  vivit.npz            transformers.VivitModel (the class vivit.py:42 instantiates), reduced size, last_hidden_state.
  hf_crosscheck.npz    HF Dinov2WithRegistersModel / SiglipVisionModel reduced-size hidden states: a cross-check for
                       the timm-semantics restatement (timm itself is not installed: "timm parity unpinned").
  prompts.json         PurePromptBuilder strings (merv/models/backbones/llm/prompting/base_prompter.py).
  preprocess.npz       per-encoder frame transforms: PIL.Image.resize (what torchvision Resize calls on a PIL image) for
                       bicubic / bilinear on 7 frame sizes, ToTensor+Normalize and the LanguageBind torch pipeline,
                       evaluated with Pillow / torch here (torchvision is absent); inputs are synthetic and regenerated.
"""
import importlib.util
import json
import math
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)


def _load(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=None)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# ----------------------------------------------------------------------------------------------------------
def gen_frame_indices():
    cases = []

    def add(N, fps, start, end, n, end_frame):
        # datasets.py:46-52
        cs, ce = start, end
        if cs is not None and math.isnan(cs):
            cs = 0
        if ce is not None and math.isnan(ce):
            ce = None
        total_secs = N / fps
        if end_frame is None or end_frame < 0:  # :131-137
            if ce is None:
                ce = total_secs
            ids = np.linspace(cs * fps, min(N - 1, ce * fps - 1), n, dtype=int)
        else:  # :138-141
            ids = np.linspace(0, min(N - 1, end_frame), n, dtype=int)
        cases.append({"N": N, "fps": fps, "clip_start_sec": start, "clip_end_sec": end, "num_frames": n,
                      "end_frame": end_frame, "ids": [int(i) for i in ids]})

    add(596, 29.97, 0.0, None, 32, 595)  # eval_data/dummy_mcq/test_q.json
    add(596, 29.97, 0.0, None, 32, None)
    add(596, 30.0, 0.0, None, 32, -1)
    add(300, 25.0, 0.0, None, 8, None)
    add(300, 25.0, 2.0, 7.5, 16, None)
    add(1000, 23.976023976023978, 1.5, 30.25, 32, None)
    add(1000, 23.976023976023978, 0.0, 100.0, 32, None)  # clip end beyond the video -> clamps to N-1
    add(20, 30.0, 0.0, None, 32, None)  # fewer frames than samples (repeats)
    add(1, 30.0, 0.0, None, 4, None)  # single frame: step == 0 branch
    add(33, 29.97, 0.0, None, 32, None)
    add(5000, 59.94, 10.0, 20.0, 32, None)
    add(5000, 59.94, float("nan"), float("nan"), 32, None)  # TVQA NaN pair
    add(5000, 59.94, 3.0, float("nan"), 16, None)
    add(240, 24.0, 0.0, None, 1, None)  # num=1
    add(240, 24.0, 0.0, None, 32, 100000)  # end_frame beyond the video
    add(240, 24.0, 0.0, None, 32, 0)
    add(240, 24.0, 5.0, None, 32, 17)
    add(12345, 29.97002997002997, 0.0, None, 32, None)
    add(901, 15.0, 0.0, 60.0, 32, None)
    add(901, 15.0, 59.0, 60.0, 32, None)
    rng = np.random.RandomState(0)
    for _ in range(60):
        N = int(rng.randint(2, 20000))
        fps = float(rng.choice([23.976023976023978, 24.0, 25.0, 29.97, 29.97002997002997, 30.0, 59.94, 60.0, 12.5]))
        total = N / fps
        s = float(rng.uniform(0, total * 0.6))
        e = float(rng.uniform(s + 0.05, total * 1.1))
        n = int(rng.choice([4, 8, 16, 32]))
        mode = rng.randint(3)
        if mode == 0:
            add(N, fps, s, e, n, None)
        elif mode == 1:
            add(N, fps, 0.0, None, n, None)
        else:
            add(N, fps, 0.0, None, n, int(rng.randint(0, N + 50)))
    (OUT / "frame_indices.json").write_text(json.dumps(cases, indent=0))
    print("frame_indices:", len(cases), "cases")


# ----------------------------------------------------------------------------------------------------------
def _stub_module(name, is_pkg=False):
    """An empty stand-in module WITH a __spec__: transformers probes optional packages through importlib.util.find_spec,
    which raises ValueError on a sys.modules entry whose __spec__ is None (the no-argument run used to die there)."""
    import importlib.machinery
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None, is_package=is_pkg)
    if is_pkg:
        m.__path__ = []
    return m


def _stub_timm():
    timm = _stub_module("timm", True)
    layers = _stub_module("timm.layers")
    models = _stub_module("timm.models", True)
    regnet = _stub_module("timm.models.regnet")
    layers.LayerNorm2d = torch.nn.LayerNorm
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    regnet.RegStage = object
    sys.modules.update({"timm": timm, "timm.layers": layers, "timm.models": models, "timm.models.regnet": regnet})


def gen_projector_fusion():
    _stub_timm()
    nn_utils = _load("ref_nn_utils", REF / "merv/util/nn_utils.py")
    torch.manual_seed(1024)  # merv.py:87 seeds with video_backbones[0].embed_dim
    out = {}
    # reduced-width projector, both spatial geometries (16->8 exact 2x2, 14->8 overlapping windows)
    for tag, S, C in (("s16", 16, 64), ("s14", 14, 48)):
        T, llm, B = 16, 128, 1
        proj = nn_utils.AveragePooling3DProjector(C, llm, output_frames=T, output_size=8, mlp_type="linear").eval()
        x = torch.randn(B, T, S * S, C)
        with torch.no_grad():
            y = proj(x)
        sd = proj.state_dict()
        out[f"proj_{tag}_keys"] = np.array(sorted(sd.keys()))
        out[f"proj_{tag}_x"] = x.numpy()
        out[f"proj_{tag}_w"] = sd["projector.projector.weight"].numpy()
        out[f"proj_{tag}_b"] = sd["projector.projector.bias"].numpy()
        out[f"proj_{tag}_y"] = y.numpy()
    # full-width single-batch slice: first 4 output tokens only (keeps the fixture small)
    proj = nn_utils.AveragePooling3DProjector(768, 4096, output_frames=16, output_size=8, mlp_type="linear").eval()
    x = torch.randn(1, 16, 196, 768)
    with torch.no_grad():
        y = proj(x)
    out["proj_full_x0"] = x[0, 0].numpy().astype(np.float16)  # frame 0 only: tokens 0..63 depend on it alone
    out["proj_full_w"] = proj.state_dict()["projector.projector.weight"][:64].numpy().astype(np.float16)
    out["proj_full_b"] = proj.state_dict()["projector.projector.bias"][:64].numpy()
    x16 = torch.from_numpy(out["proj_full_x0"].astype(np.float32))
    w16 = torch.from_numpy(out["proj_full_w"].astype(np.float32))
    proj2 = nn_utils.AveragePooling3DProjector(768, 64, output_frames=1, output_size=8, mlp_type="linear").eval()
    with torch.no_grad():
        proj2.projector.projector.weight.copy_(w16)
        proj2.projector.projector.bias.copy_(torch.from_numpy(out["proj_full_b"]))
        out["proj_full_y"] = proj2(x16[None, None]).numpy()
    # fusion (cross_attention_avg_lq): reduced and near-full embed dims
    for tag, Ed, llm, T, E, B in (("small", 96, 128, 256, 4, 2), ("e1", 96, 128, 64, 1, 2), ("wide", 384, 256, 64, 4, 1)):
        fus = nn_utils.CrossAttentionAdapterLearnableQuery(embed_dim=Ed, llm_dim=llm, token_length=T,
                                                           averagetoken=True).eval()
        with torch.no_grad():
            fus.attention.in_proj_bias.normal_(0, 0.1)
            fus.Q.mul_(8.0)  # spread the softmax away from uniform
        V = [torch.randn(B, T, llm) + 0.2 * e for e in range(E)]
        with torch.no_grad():
            y, w = fus(V)
        sd = fus.state_dict()
        out[f"fus_{tag}_keys"] = np.array(sorted(sd.keys()))
        for k in ("Q", "attention.q_proj_weight", "attention.k_proj_weight", "attention.in_proj_bias"):
            out[f"fus_{tag}_{k}"] = sd[k].numpy()
        out[f"fus_{tag}_V"] = torch.stack(V, 0).numpy()
        out[f"fus_{tag}_y"] = y.numpy()
        out[f"fus_{tag}_w"] = w.numpy()
    np.savez_compressed(OUT / "projector_fusion.npz", **out)
    print("projector_fusion: ok", {k: v.shape for k, v in out.items() if k.endswith("_y")})


# ----------------------------------------------------------------------------------------------------------
def gen_projector_upsample():
    """The reference's AveragePooling3DProjector on ONE token per frame (the registry's class-token / averaged / pooled selections reach
    MERV.forward as [B, T, 1, C], merv.py:576-585): AdaptiveAvgPool3d((T, 8, 8)) pools the 1 x 1 grid UP -- every output cell is that
    token -- before the Linear. Pins the window rule for S < out_size (round 5: such ids run through MERV) on the reference class itself."""
    _stub_timm()
    nn_utils = _load("ref_nn_utils_up", REF / "merv/util/nn_utils.py")
    torch.manual_seed(77)
    out = {}
    for tag, S, T, C, llm, B in (("s1", 1, 16, 64, 128, 2), ("s2", 2, 8, 64, 128, 1)):
        proj = nn_utils.AveragePooling3DProjector(C, llm, output_frames=T, output_size=8, mlp_type="linear").eval()
        x = torch.randn(B, T, S * S, C)
        with torch.no_grad():
            y = proj(x)
        sd = proj.state_dict()
        out[f"{tag}_x"] = x.numpy()
        out[f"{tag}_w"] = sd["projector.projector.weight"].numpy()
        out[f"{tag}_b"] = sd["projector.projector.bias"].numpy()
        out[f"{tag}_y"] = y.numpy()
    np.savez_compressed(OUT / "projector_upsample.npz", **out)
    print("projector_upsample: ok", {k: v.shape for k, v in out.items() if k.endswith("_y")})


# ----------------------------------------------------------------------------------------------------------
def _load_languagebind():
    import transformers.models.clip.modeling_clip as mc
    if not hasattr(mc, "clip_loss"):
        mc.clip_loss = lambda similarity: similarity.mean()  # never called on the vision path
    peft = _stub_module("peft")
    peft.LoraConfig = object
    peft.get_peft_model = lambda m, c: m
    sys.modules["peft"] = peft
    pkg = types.ModuleType("ref_lbvideo")
    pkg.__path__ = [str(REF / "merv/models/backbones/video/languagebind/video")]
    sys.modules["ref_lbvideo"] = pkg
    base = REF / "merv/models/backbones/video/languagebind/video"
    conf = _load("ref_lbvideo.configuration_video", base / "configuration_video.py", package="ref_lbvideo")
    mod = _load("ref_lbvideo.modeling_video", base / "modeling_video.py", package="ref_lbvideo")
    return conf, mod


def gen_languagebind():
    conf, mod = _load_languagebind()
    out = {}
    for tag, act in (("gelu", "gelu"), ("quick", "quick_gelu")):
        torch.manual_seed(7)
        cfg = conf.CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                                    image_size=56, patch_size=14, hidden_act=act, layer_norm_eps=1e-5,
                                    add_time_attn=True, num_frames=8)
        cfg._attn_implementation = "eager"
        vt = mod.CLIPVisionTransformer(cfg).eval()
        with torch.no_grad():
            for p in vt.parameters():  # spread everything away from the trivial init
                if p.dim() == 1:
                    p.add_(torch.randn_like(p) * 0.1)
            pix = torch.randn(2, 3, 16, 56, 56)  # [B, C, T, H, W], T=16 with num_frames=8 => two clips per video
            o = vt(pix, output_hidden_states=True, return_dict=True)
        hs = o.hidden_states[-2]  # languagebind/__init__.py:85
        out[f"{tag}_pix"] = pix.numpy()
        out[f"{tag}_hs_m2"] = hs.numpy()  # [B, T, 17, 128]
        for k, v in vt.state_dict().items():
            out[f"{tag}_sd/{k}"] = v.numpy()
    np.savez_compressed(OUT / "languagebind.npz", **out)
    print("languagebind: ok", out["gelu_hs_m2"].shape)


def gen_vivit():
    from transformers import VivitConfig, VivitModel
    torch.manual_seed(11)
    cfg = VivitConfig(image_size=64, num_frames=8, tubelet_size=[2, 16, 16], hidden_size=128, num_hidden_layers=2,
                      num_attention_heads=2, intermediate_size=256)
    assert cfg.hidden_act == "gelu_fast" and cfg.layer_norm_eps == 1e-6 and cfg.qkv_bias
    m = VivitModel(cfg, add_pooling_layer=False).eval()
    out = {}
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
        m.embeddings.cls_token.normal_(0, 0.5)
        m.embeddings.position_embeddings.normal_(0, 0.5)
        pix = torch.randn(2, 8, 3, 64, 64)
        y = m(pix).last_hidden_state
    out["pix"] = pix.numpy()
    out["last_hidden_state"] = y.numpy()
    for k, v in m.state_dict().items():
        out[f"sd/{k}"] = v.numpy()
    np.savez_compressed(OUT / "vivit.npz", **out)
    print("vivit: ok", y.shape)


def gen_hf_crosscheck():
    out = {}
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    torch.manual_seed(13)
    cfg = Dinov2WithRegistersConfig(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, mlp_ratio=2, image_size=56,
                                    patch_size=14, num_register_tokens=4, layerscale_value=1.0, hidden_act="gelu",
                                    layer_norm_eps=1e-6, qkv_bias=True)
    m = Dinov2WithRegistersModel(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
        m.embeddings.cls_token.normal_(0, 0.5)
        m.embeddings.register_tokens.normal_(0, 0.5)
        m.embeddings.position_embeddings.normal_(0, 0.5)
        pix = torch.randn(2, 3, 56, 56)
        o = m(pix, output_hidden_states=True)
    out["dino_pix"] = pix.numpy()
    out["dino_hs"] = torch.stack(o.hidden_states, 0).numpy()  # [L+1, B, 1+4+16, D]
    for k, v in m.state_dict().items():
        out[f"dino_sd/{k}"] = v.numpy()
    from transformers import SiglipVisionConfig, SiglipVisionModel
    cfg = SiglipVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                             image_size=64, patch_size=16, hidden_act="gelu", layer_norm_eps=1e-6)
    m = SiglipVisionModel(cfg).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
        pix = torch.randn(2, 3, 64, 64)
        o = m(pix, output_hidden_states=True)
    out["siglip_pix"] = pix.numpy()
    out["siglip_hs"] = torch.stack(o.hidden_states, 0).numpy()
    for k, v in m.state_dict().items():
        if "head" in k:
            continue
        out[f"siglip_sd/{k}"] = v.numpy()
    np.savez_compressed(OUT / "hf_crosscheck.npz", **out)
    print("hf_crosscheck: ok", out["dino_hs"].shape, out["siglip_hs"].shape)


def gen_prompts():
    base = REF / "merv/models/backbones/llm/prompting"
    pkg = types.ModuleType("ref_prompting")
    pkg.__path__ = [str(base)]
    sys.modules["ref_prompting"] = pkg
    bp = _load("ref_prompting.base_prompter", base / "base_prompter.py", package="ref_prompting")
    cases = []
    for turns in (["What is happening in this video?"], ["Describe the video.", "A cat jumps.", "What colour is it?"]):
        pb = bp.PurePromptBuilder("merv")
        seq = []
        for i, msg in enumerate(turns):
            role = "human" if i % 2 == 0 else "gpt"
            ret = pb.add_turn(role, msg)
            seq.append({"role": role, "message": msg, "wrapped": ret})
        cases.append({"turns": seq, "prompt": pb.get_prompt(),
                      "potential": pb.get_potential_prompt("And then?") if len(turns) % 2 == 0 else None})
    (OUT / "prompts.json").write_text(json.dumps(cases, indent=1))
    print("prompts: ok", cases[0]["prompt"])
    # chat builders: their files import the base class through the absolute package path
    chain = "merv.models.backbones.llm.prompting.base_prompter".split(".")
    for i in range(1, len(chain)):
        name = ".".join(chain[:i])
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules[".".join(chain)] = bp
    chat_cases = []
    for fname, cls in (("llama2_chat_prompter.py", "LLaMa2ChatPromptBuilder"), ("vicuna_v15_prompter.py", "VicunaV15ChatPromptBuilder")):
        mod = _load("ref_prompting." + fname[:-3], base / fname, package="ref_prompting")
        for sysprompt in (None, "Answer briefly."):
            for turns in (["<image>\nWhat is happening in this video?"], ["Describe the video.", "A cat jumps.", "What colour is it?"],
                          ["Q1", "", "Q2", "A2"]):
                pb = getattr(mod, cls)("merv", system_prompt=sysprompt)
                first_potential = pb.get_potential_prompt("And then?")
                seq = []
                for i, msg in enumerate(turns):
                    role = "human" if i % 2 == 0 else "gpt"
                    seq.append({"role": role, "message": msg, "wrapped": pb.add_turn(role, msg)})
                chat_cases.append({"builder": cls, "system_prompt": sysprompt, "turns": seq, "prompt": pb.get_prompt(),
                                   "first_potential": first_potential,
                                   "potential": pb.get_potential_prompt("And then?") if len(turns) % 2 == 0 else None})
    (OUT / "prompts_chat.json").write_text(json.dumps(chat_cases, indent=1))
    print("prompts_chat: ok", len(chat_cases))
    # the header-style builders (Llama-3.1 in llama2_chat_prompter.py, Qwen2): no system-prompt argument, no stripping
    hdr_cases = []
    l2 = sys.modules["ref_prompting.llama2_chat_prompter"]
    q2 = _load("ref_prompting.qwen2_prompter", base / "qwen2_prompter.py", package="ref_prompting")
    for mod, cls in ((l2, "LLaMa31PromptBuilder"), (q2, "Qwen2PromptBuilder")):
        for turns in (["<image>\nWhat is happening in this video?"], ["Describe the video. ", "A cat jumps.", "What colour is it?"], ["Q1", "", "Q2", "A2"]):
            pb = getattr(mod, cls)("merv")
            seq = []
            for i, msg in enumerate(turns):
                role = "human" if i % 2 == 0 else "gpt"
                seq.append({"role": role, "message": msg, "wrapped": pb.add_turn(role, msg)})
            hdr_cases.append({"builder": cls, "turns": seq, "prompt": pb.get_prompt()})
    (OUT / "prompts_header.json").write_text(json.dumps(hdr_cases, indent=1))
    print("prompts_header: ok", len(hdr_cases))




# ----------------------------------------------------------------------------------------------------------
def synth_frame(H, W, seed):
    """Deterministic integer-only test frame [H,W,3] uint8 (regenerated identically by the tests)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    base = ((xx * 3 + yy * 5 + (xx * yy) % 7 * 9) % 256)[..., None] + np.array([0, 17, 34])
    noise = rng.randint(0, 64, size=(H, W, 3))
    return ((base + noise) % 256).astype(np.uint8)


PRE_SIZES = [(96, 128), (300, 200), (231, 487), (224, 224), (224, 100), (50, 224), (720, 1280)]
LB_SIZES = [(96, 128), (300, 200)]


def gen_preprocess():
    """Per-encoder CPU transforms at the level the reference reaches them: torchvision is not installed here, so
    Resize((224,224)) on a PIL image is evaluated as what it calls -- PIL.Image.resize((224,224), resample) -- and
    ToTensor / Normalize / NormalizeVideo / ShortSideScale / CenterCropVideo as the torch ops they are made of
    (dinov2_video.py:94-124, siglip.py:104-134, vivit.py:54-92, processing_video.py:28-79). Inputs are synthetic and
    regenerated by the tests from (H, W, seed); only outputs are stored."""
    from PIL import Image
    out = {"sizes": np.array(PRE_SIZES), "lb_sizes": np.array(LB_SIZES)}
    for i, (H, W) in enumerate(PRE_SIZES):
        pil = Image.fromarray(synth_frame(H, W, 100 + i), "RGB")
        for name, flt in (("bicubic", Image.BICUBIC), ("bilinear", Image.BILINEAR)):
            out[f"img{i}_{name}_u8"] = np.asarray(pil.resize((224, 224), flt)).transpose(2, 0, 1).copy()
    # float stages: ToTensor + Normalize (fp32 torch ops) on one resized image, stored as a thin slice
    t = torch.from_numpy(out["img1_bicubic_u8"].copy()).to(torch.float32).div(255)
    mean = torch.tensor((0.485, 0.456, 0.406)).view(3, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225)).view(3, 1, 1)
    out["img1_dinov2_pix_rows"] = t.clone().sub_(mean).div_(std)[:, ::16].numpy()
    # LanguageBind pipeline (flip off) on 2-frame clips
    for j, (H, W) in enumerate(LB_SIZES):
        frames = np.stack([synth_frame(H, W, 200 + 10 * j + f) for f in range(2)], 0)  # [T, H, W, 3]
        clip = torch.from_numpy(frames).permute(3, 0, 1, 2)  # [C, T, H, W] == video.permute(1,0,2,3) of [T,C,H,W]
        x = clip / 255.0
        m = torch.tensor((0.48145466, 0.4578275, 0.40821073)); sd = torch.tensor((0.26862954, 0.26130258, 0.27577711))
        x = x.clone().sub_(m[:, None, None, None]).div_(sd[:, None, None, None])
        c, t_, h, w = x.shape
        if w < h:
            new_h, new_w = int(math.floor((float(h) / w) * 224)), 224
        else:
            new_h, new_w = 224, int(math.floor((float(w) / h) * 224))
        x = torch.nn.functional.interpolate(x, size=(new_h, new_w), mode="bilinear", align_corners=False)
        i0, j0 = int(round((new_h - 224) / 2.0)), int(round((new_w - 224) / 2.0))
        out[f"lb{j}_out"] = x[..., i0:i0 + 224, j0:j0 + 224].numpy().astype(np.float32)
    np.savez_compressed(OUT / "preprocess.npz", **out)
    print("preprocess: ok", len(out), "arrays")




# ----------------------------------------------------------------------------------------------------------
def _ref_method(path: Path, cls: str, name: str, extra_globals=None):
    """The function object of `cls.name` defined in the reference file `path`, compiled from that file's own AST node
    (importing the module would need timm / torchvision / the hub; the method bodies need only torch)."""
    import ast
    tree = ast.parse(path.read_text())
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    fn = next(n for n in node.body if isinstance(n, ast.FunctionDef) and n.name == name)
    fn.decorator_list = []
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"torch": torch, "Tuple": tuple, "Callable": object}
    ns.update(extra_globals or {})
    exec(compile(ast.fix_missing_locations(mod), str(path), "exec"), ns)
    return ns[name]


def gen_token_selection():
    """The registry's token selections (materialize.py:31-73): the reference's own VideoBackbone.forward / num_patches bodies
    (languagebind/__init__.py:79-130, dinov2_video.py:132-170, vivit.py:100-150) run on small random hidden states through a
    stand-in `self` whose featurizer returns them. Pins oracle.select_tokens and merv_amd.backbones' selections."""
    VB = REF / "merv/models/backbones/video"
    NS = types.SimpleNamespace
    g = torch.Generator().manual_seed(321)
    out, meta = {}, {}
    B, D = 2, 2
    # LanguageBind: hidden_states[-2] is [B, F, 257, D]
    F_lb = 4
    hid = torch.randn(B, F_lb, 257, D, generator=g)
    out["languagebind_hidden"] = hid.numpy()
    fwd = _ref_method(VB / "languagebind/__init__.py", "LangBindVideoBackbone", "forward")
    npatch = _ref_method(VB / "languagebind/__init__.py", "LangBindVideoBackbone", "num_patches")
    for token in (None, "average", "classemb", "noclass", "classemb-at-first"):
        me = NS(featurizer=lambda v, output_hidden_states: NS(hidden_states=[None, hid, None]), token=token, embed_dim=D, num_frames=F_lb)
        y = fwd(me, torch.zeros(B, 3, F_lb, 1, 1), None)
        out[f"languagebind_{token}"] = y.numpy()
        meta[f"languagebind_{token}"] = {"num_patches": int(npatch(me))}
    # DINOv2: the featurizer returns what dinov2_video.py:46-66 patched it to return for the id
    F_d, P = 3, 256
    hidden = torch.randn(B * F_d, 5 + P, D, generator=g)  # layer L-2, prefix tokens first
    pooled = torch.randn(B * F_d, D, generator=g)         # timm forward(): final norm + class token (bare id)
    out["dinov2_hidden"], out["dinov2_pooled"] = hidden.numpy(), pooled.numpy()
    fwd = _ref_method(VB / "dinov2_video.py", "DinoV2VideoBackbone", "forward")
    npatch = _ref_method(VB / "dinov2_video.py", "DinoV2VideoBackbone", "num_patches")
    for ident in ("dinov2-video", "dinov2-video-all-tokens", "dinov2-video-all-token-with-cls", "dinov2-video-classemb-at-first"):
        if "all-token-with-cls" in ident or "classemb-at-first" in ident:
            ret = (hidden[:, 5:], hidden[:, :5])  # get_intermediate_layers(return_prefix_tokens=True) through unpack_tuple
        elif "all-token" in ident:
            ret = hidden[:, 5:]
        else:
            ret = pooled
        me = NS(featurizer=lambda v, r=ret: r, identifier=ident, embed_dim=D, num_frames=F_d,
                )
        me.featurizer.patch_embed = NS(num_patches=P)
        y = fwd(me, torch.zeros(B, F_d, 3, 1, 1), None)
        out[ident] = y.numpy()
        meta[ident] = {"num_patches": int(npatch(me))}
    # ViViT: last_hidden_state [B, 3137, D]
    last = torch.randn(B, 3137, D, generator=g)
    out["vivit_hidden"] = last.numpy()
    fwd = _ref_method(VB / "vivit.py", "ViVITVideoBackbone", "forward")
    npatch = _ref_method(VB / "vivit.py", "ViVITVideoBackbone", "num_patches")
    for ident in ("vivit-google-b-cls-token", "vivit-google-b-all-tokens", "vivit-google-b-all-no-cls",
                  "vivit-google-b-all-no-cls-16frames", "vivit-google-b-classemb-at-first-16frames"):
        me = NS(featurizer=lambda v: NS(last_hidden_state=last), video_backbone_id=ident,
                huggingface_path_or_url="google/vivit-b-16x2-kinetics400")
        y = fwd(me, torch.zeros(B, 32, 3, 1, 1), None)
        out[ident] = y.numpy()
        meta[ident] = {"num_patches": int(npatch(me))}
    np.savez_compressed(OUT / "token_selection.npz", **out)
    (OUT / "token_selection.json").write_text(json.dumps(meta, indent=1))
    print("token_selection:", {k: tuple(v.shape) for k, v in out.items() if "hidden" not in k and "pooled" not in k})




# ----------------------------------------------------------------------------------------------------------
def gen_siglip_pool():
    """The attention-pooling head behind the SigLIP ids that keep timm's forward() (siglip.py:46-63: every id without
    `all-no-cls`): timm's AttentionPoolLatent is absent here, so the oracle's restatement is pinned on the same head as
    transformers implements it (SiglipMultiheadAttentionPoolingHead: a learnt probe attends over the final-norm tokens,
    then x + mlp(norm(x)), token 0) -- last_hidden_state in, pooler_output out, the head's state dict."""
    from transformers import SiglipVisionConfig, SiglipVisionModel
    torch.manual_seed(29)
    cfg = SiglipVisionConfig(hidden_size=128, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2,
                             image_size=64, patch_size=16, hidden_act="gelu", layer_norm_eps=1e-6)
    m = SiglipVisionModel(cfg).eval()
    out = {}
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "head" in n:
                p.add_(torch.randn_like(p) * (0.1 if p.dim() == 1 else 0.05))
        o = m(torch.randn(3, 3, 64, 64))
    out["last_hidden_state"] = o.last_hidden_state.numpy()   # [3, 16, 128], after post_layernorm
    out["pooler_output"] = o.pooler_output.numpy()           # [3, 128]
    for k, v in m.state_dict().items():
        if "head." in k:  # `head.*` (transformers 5) or `vision_model.head.*` (4.x)
            out["sd/" + k[k.index("head.") + 5:]] = v.numpy()
    np.savez_compressed(OUT / "siglip_pool.npz", **out)
    print("siglip_pool: ok", out["pooler_output"].shape, sorted(k for k in out if k.startswith("sd/")))




# ----------------------------------------------------------------------------------------------------------
def _ref_class(path: Path, cls: str, names, base, extra_globals=None):
    """A class object holding the reference's own method bodies `names` of `cls` (compiled from that file's AST nodes,
    decorators kept) on top of the stand-in base class `base`, so `self.<collaborator>` and a zero-argument `super()`
    resolve to stand-ins. The module cannot be imported here (draccus / timm / torchvision); the bodies need only torch.
    Compiled with `from __future__ import annotations` in force, as the reference file declares (merv.py:14)."""
    import __future__
    import ast
    tree = ast.parse(path.read_text())
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    fns = [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names), (names, [f.name for f in fns])
    klass = ast.ClassDef(name=cls, bases=[ast.Name(id="_StandInBase", ctx=ast.Load())], keywords=[], body=fns, decorator_list=[])
    if sys.version_info >= (3, 12):
        klass.type_params = []
    ns = {"torch": torch, "np": np, "_StandInBase": base}
    ns.update(extra_globals or {})
    code = compile(ast.fix_missing_locations(ast.Module(body=[klass], type_ignores=[])), str(path), "exec",
                   flags=__future__.annotations.compiler_flag)
    exec(code, ns)
    return ns[cls], ns  # ns is the methods' globals: stand-ins for module-level names (load_video, ...) go there


def gen_merv_forward():
    """The reference's own `MERV.forward` body (merv/models/vidlms/merv.py:503-734) and the frame sub-sampling of
    `MERV.generate` (:778-830, the `video[:: max(num_frames) // num_frame]` of :803-806) executed here on a stand-in `self`:
    video backbones are callables returning given tokens (with temporal_resolution / spatial_resolution), projectors and
    fusion are the reference's real nn_utils classes, `llm_backbone` records what it is called with. Pins the oracle's
    visual_path tail (reshape -> projectors -> fusion), splice and assemble_training_batch, and the product's
    merv_splice_forward / train.assemble_training_batch / sampler.temporal_subsample."""
    from typing import List, Optional
    _stub_timm()
    nn_utils = _load("ref_nn_utils", REF / "merv/util/nn_utils.py")
    NS = types.SimpleNamespace
    out, meta = {}, {}

    class Recorder:
        def __init__(self, table, bos_token):
            self.table, self.tokenizer, self.calls = table, NS(bos_token=bos_token, pad_token_id=0), []
            self.half_precision_dtype = torch.bfloat16

        def embed_input_ids(self, ids):
            return self.table[ids]

        def __call__(self, **kw):
            self.calls.append(kw)
            return kw

    class Base:
        def generate(self, **kw):  # GenerationMixin.generate stand-in (merv.py:818)
            self.generate_kwargs = kw
            return torch.zeros(1, kw["input_ids"].shape[1] + 3, dtype=torch.long)

    MERV, ref_globals = _ref_class(REF / "merv/models/vidlms/merv.py", "MERV", ["forward", "generate"], Base,
                                   {"IGNORE_INDEX": -100, "Optional": Optional, "List": List, "CausalLMOutputWithPast": dict,
                                    "Image": None, "load_video": None})

    # ---- forward: merv-full's structure at reduced widths: 4 encoders, T = 2 frames, spatial 16^2 / 16^2 / 14^2 / 14^2 -> 8^2,
    #      so 128 visual tokens of width llm = 128 (widths the HIP GEMM accepts: K % 64 == 0, N % 128 == 0, so the SAME fixture
    #      drives the -m gpu test; encoder tokens fp16-representable: stored as fp16)
    g = torch.Generator().manual_seed(2024)
    T, llm, Cs, Ss = 2, 128, (64, 64, 64, 64), (256, 256, 196, 196)
    torch.manual_seed(1024)  # merv.py:87
    projectors = [nn_utils.AveragePooling3DProjector(C, llm, output_frames=T, output_size=8, mlp_type="linear").eval() for C in Cs]
    fusion = nn_utils.CrossAttentionAdapterLearnableQuery(embed_dim=48, llm_dim=llm, token_length=T * 64, averagetoken=True).eval()
    with torch.no_grad():
        fusion.attention.in_proj_bias.normal_(0, 0.1, generator=g)
        fusion.Q.mul_(8.0)
    for i, p in enumerate(projectors):
        out[f"proj{i}_w"] = p.state_dict()["projector.projector.weight"].numpy()
        out[f"proj{i}_b"] = p.state_dict()["projector.projector.bias"].numpy()
    for k in ("Q", "attention.q_proj_weight", "attention.k_proj_weight", "attention.in_proj_bias"):
        out[f"fus_{k}"] = fusion.state_dict()[k].numpy()
    vocab = 50
    table = torch.randn(vocab, llm, generator=g)
    out["embed_table"] = table.numpy()

    def run(tag, B, S, multimodal_indices, bos_token, with_masks=True):
        feats = [torch.randn(B, T * s, C, generator=g).half().float() for s, C in zip(Ss, Cs)]
        fused_rec = {}

        def fusion_rec(xs):
            y, w = fusion(xs)
            fused_rec["y"], fused_rec["w"] = y, w
            return y, w

        me = MERV.__new__(MERV)
        me.llm_backbone = Recorder(table, bos_token)
        me.video_backbone_requires_grad = False
        me.video_backbones = []
        for f, s in zip(feats, Ss):  # a backbone: called as vb(video_values[i], is_image) (merv.py:564), returns its tokens
            me.video_backbones.append(type("VB", (), {"temporal_resolution": T, "spatial_resolution": s,
                                                      "__call__": staticmethod((lambda f_: (lambda video, is_image: f_))(f))})())
        me.tokens_resampled = True
        me.projectors = projectors
        me.feature_fusion_type = "cross_attention_avg_lq"
        me.feature_fusion = fusion_rec
        input_ids = torch.randint(2, vocab, (B, S), generator=g)
        input_ids[:, 0] = 1
        am = torch.ones(B, S, dtype=torch.bool)
        lab = input_ids.clone()
        lab[:, : S // 2] = -100
        for b in range(B):  # ragged right padding
            pad = b % 3
            if pad:
                am[b, S - pad:] = False
                lab[b, S - pad:] = -100
        with torch.no_grad():
            ret = me.forward(input_ids=input_ids, attention_mask=am if with_masks else None,
                             video_values=[torch.zeros(B, 1)] * 4, labels=lab if with_masks else None,
                             multimodal_indices=multimodal_indices)
        assert ret is me.llm_backbone.calls[-1] and ret["input_ids"] is None
        for i, f in enumerate(feats):
            out[f"{tag}_feat{i}"] = f.numpy().astype(np.float16)
        out[f"{tag}_input_ids"] = input_ids.numpy()
        out[f"{tag}_attention_mask"] = am.numpy()
        out[f"{tag}_labels"] = lab.numpy()
        if multimodal_indices is not None:
            out[f"{tag}_multimodal_indices"] = multimodal_indices.numpy()
        out[f"{tag}_fused"] = fused_rec["y"].numpy()
        out[f"{tag}_fusion_weights"] = fused_rec["w"].numpy()
        out[f"{tag}_inputs_embeds"] = ret["inputs_embeds"].numpy()
        if with_masks:
            out[f"{tag}_out_attention_mask"] = ret["attention_mask"].numpy()
            out[f"{tag}_out_labels"] = ret["labels"].numpy()
        else:
            assert ret["attention_mask"] is None and ret["labels"] is None
        meta[tag] = {"B": B, "S": S, "bos_token_length": 1 if bos_token is not None else 0, "with_masks": with_masks,
                     "multimodal_indices": None if multimodal_indices is None else multimodal_indices.tolist()}

    run("full", 2, 9, None, "<s>")                                   # fully multimodal batch (multimodal_indices None, :543-545)
    run("mixed", 4, 7, torch.tensor([0, 2, 3]), "<s>")               # multimodal rows first, unimodal rows padded at the end (:666-719)
    run("nobos", 1, 6, None, None)                                   # tokenizer without BOS (Qwen2.5; :520-521)
    run("infer", 1, 8, None, "<s>", with_masks=False)                # generate()-style call: no mask, no labels
    meta["geometry"] = {"T": T, "llm": llm, "C": list(Cs), "S": list(Ss), "out_size": 8, "embed_dim": 48}

    # ---- generate(): what reaches `video_values` for a given num_frames list (merv.py:796-806)
    sub = []
    for num_frames, n_loaded in (([16, 16, 32, 16], 32), ([8, 8, 32, 8], 32), ([16, 16, 32, 12], 32), ([4], 4), ([5, 32, 7], 32), ([16, 16, 32, 16], 20)):
        seen = {}

        def fake_load_video(video, clip_start_sec=0.0, clip_end_sec=None, num_frames=None, end_frame=None):
            seen.update(num_frames=num_frames, clip_start_sec=clip_start_sec, clip_end_sec=clip_end_sec, end_frame=end_frame)
            return torch.arange(n_loaded).view(n_loaded, 1, 1, 1).expand(n_loaded, 3, 2, 2)  # frame k holds the value k

        ref_globals["load_video"] = fake_load_video
        me = MERV.__new__(MERV)
        me.device = torch.device("cpu")
        me.enable_mixed_precision_training = False
        tok = lambda text, truncation, return_tensors: NS(input_ids=torch.tensor([[1, 5, 6]]))  # noqa: E731
        tok.pad_token_id = 0
        tok.decode = lambda ids, skip_special_tokens: " ok "
        me.llm_backbone = NS(tokenizer=tok, half_precision_dtype=torch.bfloat16)
        me.video_backbones = [NS(video_transform=(lambda v: v.float()), default_video_resolution=(1,)) for _ in num_frames]
        text = me.generate("clip.mp4", "prompt", num_frames, end_frame=17, do_sample=False)
        assert text == "ok"
        kw = me.generate_kwargs
        sub.append({"num_frames": num_frames, "frames_loaded": n_loaded, "load_video_num_frames": seen["num_frames"],
                    "end_frame_forwarded": seen["end_frame"],
                    "selected": [v[0, :, 0, 0, 0].long().tolist() for v in kw["video_values"]],
                    "forwarded_kwargs": sorted(k for k in kw if k not in ("video_values",))})
    meta["generate_subsample"] = sub
    np.savez_compressed(OUT / "merv_forward.npz", **out)
    (OUT / "merv_forward.json").write_text(json.dumps(meta, indent=1))
    print("merv_forward: ok", {k: tuple(v.shape) for k, v in out.items() if k.endswith("inputs_embeds")},
          [s["selected"][-1][:4] for s in sub])


GENERATORS = {"frames": gen_frame_indices, "projfus": gen_projector_fusion, "lb": gen_languagebind, "vivit": gen_vivit,
              "hf": gen_hf_crosscheck, "prompts": gen_prompts, "preprocess": gen_preprocess,
              "token_selection": gen_token_selection, "siglip_pool": gen_siglip_pool, "merv_forward": gen_merv_forward,
              "projector_upsample": gen_projector_upsample}


def main(argv):
    """python3 tools/make_goldens.py [--out DIR] [generator ...]   (no generator names: all of them, in this order)"""
    global OUT
    argv = list(argv)
    if "--out" in argv:
        i = argv.index("--out")
        OUT = Path(argv[i + 1])
        OUT.mkdir(parents=True, exist_ok=True)
        del argv[i:i + 2]
    unknown = [a for a in argv if a not in GENERATORS]
    if unknown:
        raise SystemExit(f"unknown generator(s) {unknown}; known: {sorted(GENERATORS)}")
    torch.set_num_threads(8)
    for name in (argv or list(GENERATORS)):
        GENERATORS[name]()


if __name__ == "__main__":
    main(sys.argv[1:])
