"""CPU: the C-ABI library loads and exports every symbol include/merv_hip.h declares; the host-only entry points
(frame-index sampler, temporal subsample) are bit-exact against the goldens; host-side logic and error behaviour.
No compute calls that need a GPU."""
import ctypes as C
import json
import re
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
G = ROOT / "tests" / "golden"


def _declared_symbols():
    text = (ROOT / "include" / "merv_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(merv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from merv_amd import _lib
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libmerv_hip.so does not export {n}"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert lib.merv_abi_version() == 4


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from merv_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("MERV_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_has_no_oracle_import():
    for p in (ROOT / "merv_amd").rglob("*.py"):
        src = p.read_text()
        assert "import oracle" not in src and "from oracle" not in src, p


def test_sampler_bit_exact_through_c_abi():
    from merv_amd.sampler import frame_indices
    cases = json.loads((G / "frame_indices.json").read_text())
    for c in cases:
        got = frame_indices(c["N"], c["fps"], c["clip_start_sec"], c["clip_end_sec"], c["num_frames"], c["end_frame"])
        assert got == c["ids"], c


def test_sampler_errors():
    from merv_amd.sampler import frame_indices, temporal_subsample
    with pytest.raises(ValueError):
        frame_indices(0, 30.0, 0.0, None, 8, None)
    assert temporal_subsample(32, 32, 16) == list(range(0, 32, 2))
    assert len(temporal_subsample(32, 32, 12)) == 16  # reference over-sampling quirk kept
    with pytest.raises(ValueError):
        temporal_subsample(32, 16, 32)  # step 0, like Python's "slice step cannot be zero"


def test_specs_match_survey_flop_table():
    from merv_amd.encoder import merv_full_specs
    want = {"languagebind": 3.281, "dinov2": 2.525, "vivit": 0.903, "siglip": 0.513}  # TFLOP, SURVEY.md section 8a
    for s in merv_full_specs():
        assert abs(s.flops_per_video() / 1e12 - want[s.name]) < 2e-3, s.name
        assert s.num_patches == {"languagebind": 4096, "dinov2": 4096, "vivit": 3136, "siglip": 3136}[s.name]
        assert s.t_out == 16
    assert merv_full_specs()[0].k_pad == 640 and merv_full_specs()[2].k_pad == 1536


def test_encoder_refuses_cpu_device():
    from merv_amd.encoder import HipEncoder, merv_full_specs
    with pytest.raises(RuntimeError, match="no CPU path"):
        HipEncoder(merv_full_specs()[3], {}, torch.device("cpu"))


def test_create_rejects_bad_descriptor():
    from merv_amd import _lib
    lib = _lib.load()
    d = _lib.EncoderDesc(dim=100, heads=2, mlp_dim=256, layers=0, patch=14, tubelet=1, img=224, frames=16)
    w = _lib.EncoderWeights()
    h = C.c_void_p()
    rc = lib.merv_encoder_create(C.byref(d), C.byref(w), C.byref(h))
    assert rc == 1 and b"dim" in lib.merv_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc, "merv_encoder_create")


def test_plan_units_properties():
    from merv_amd.distributed import atom_frames
    from merv_amd.encoder import merv_full_specs
    from merv_amd.visual_path import plan_one_encoder_per_rank, plan_units
    specs = merv_full_specs()
    costs = [s.flops_per_video() for s in specs]
    frames = [s.frames for s in specs]
    atoms = [atom_frames(s) for s in specs]
    assert atoms == [8, 1, 32, 1]  # LanguageBind clips, DINOv2 frames, ViViT whole, SigLIP frames
    for world in (1, 2, 4, 8):
        for per in (1, 2, 8):
            Gv = world * per
            plan = plan_units(costs, Gv, world, frames, atoms)
            seen = set()
            for units in plan:
                for (e, v0, v1, f0, f1) in units:
                    assert 0 <= v0 < v1 <= Gv and 0 <= f0 < f1 <= frames[e] and f0 % atoms[e] == 0 and f1 % atoms[e] == 0
                    for v in range(v0, v1):
                        for f in range(f0, f1):
                            assert (e, v, f) not in seen
                            seen.add((e, v, f))
            assert len(seen) == Gv * sum(frames)
            load = [sum(costs[e] * (v1 - v0) * (f1 - f0) / frames[e] for (e, v0, v1, f0, f1) in u) for u in plan]
            if per >= 8:
                assert max(load) / (sum(load) / world) < 1.02, (world, per, load)
    # whole videos only (no atoms given): units never split a video
    for units in plan_units(costs, 5, 2):
        assert all((f0, f1) == (0, 1) for (_, _, _, f0, f1) in units)
    # north_star's literal placement: 4 GPUs -> one encoder per GPU; the balanced plan of the same case splits DINOv2 by frames
    p = plan_one_encoder_per_rank(4, 1, 4, frames)
    assert [u[0][0] for u in p] == [0, 1, 2, 3] and all(len(u) == 1 for u in p)
    bal = plan_units(costs, 1, 4, frames, atoms)
    assert any(e == 1 and (f0, f1) != (0, 16) for units in bal for (e, _, _, f0, f1) in units)
    assert not any(e == 2 and (f0, f1) != (0, 32) for units in bal for (e, _, _, f0, f1) in units)  # ViViT never splits


def test_registry_keys_and_unwired_variants():
    from merv_amd.backbones import VIDEO_BACKBONES, get_video_backbone_and_transform
    for k in ("languagebind-video-noclass", "dinov2-video-all-tokens", "vivit-google-b-all-no-cls-16frames",
              "siglip-vit-b16-224px-all-no-cls", "dinov2-video", "languagebind-video-classemb"):
        assert k in VIDEO_BACKBONES
    with pytest.raises(ValueError, match="is not supported"):
        get_video_backbone_and_transform(["no-such-backbone"], "resize-naive", [8])
    # the one key that cannot be wired: it fails inside the reference itself
    k = "siglip-vit-b16-224px-classemb-at-first"
    with pytest.raises(NotImplementedError, match="fails inside the reference"):
        VIDEO_BACKBONES[k]["cls"](k, "resize-naive", num_frames=8, weights="random", device="cpu", **VIDEO_BACKBONES[k]["kwargs"])
    with pytest.raises(ValueError, match="does not exist"):
        VIDEO_BACKBONES["languagebind-video"]["cls"]("languagebind-video", "resize-naive", num_frames=8, token="first", weights="random")
    with pytest.raises(ValueError, match="no hub access"):
        VIDEO_BACKBONES["siglip-vit-b16-224px-all-no-cls"]["cls"]("siglip-vit-b16-224px-all-no-cls", "resize-naive", num_frames=8)


def test_mervvisual_rejects_unsupported_arch():
    from merv_amd.vidlm import MERVVisual

    class FakeBB:
        embed_dim, temporal_resolution = 64, 16
    with pytest.raises(ValueError):
        MERVVisual([FakeBB()], arch_specifier="gelu-mlp")
    with pytest.raises(ValueError):
        MERVVisual([FakeBB()], arch_specifier="avg+linear")


def test_prompt_builders_match_reference_strings():
    from merv_amd.prompting import LLaMa2ChatPromptBuilder, PurePromptBuilder
    cases = json.loads((G / "prompts.json").read_text())  # produced by the reference's PurePromptBuilder
    for c in cases:
        pb = PurePromptBuilder("merv")
        for t in c["turns"]:
            assert pb.add_turn(t["role"], t["message"]) == t["wrapped"]
        assert pb.get_prompt() == c["prompt"]
        if c["potential"] is not None:
            assert pb.get_potential_prompt("And then?") == c["potential"]
    chat = LLaMa2ChatPromptBuilder("merv")
    w = chat.add_turn("human", "<image>\nHi")
    assert w.startswith("<s>[INST] <<SYS>\n") and w.endswith("Hi [/INST] ") and not chat.get_prompt().startswith("<s>")
    with pytest.raises(AssertionError):
        chat.add_turn("human", "again")  # roles must alternate


def test_chat_prompt_builders_match_reference_strings():
    import merv_amd.prompting as P
    cases = json.loads((G / "prompts_chat.json").read_text())  # reference LLaMa2Chat / VicunaV15 builders (tools/make_goldens.py)
    assert {c["builder"] for c in cases} == {"LLaMa2ChatPromptBuilder", "VicunaV15ChatPromptBuilder"}
    for c in cases:
        pb = getattr(P, c["builder"])("merv", system_prompt=c["system_prompt"])
        assert pb.get_potential_prompt("And then?") == c["first_potential"]
        for t in c["turns"]:
            assert pb.add_turn(t["role"], t["message"]) == t["wrapped"]
        assert pb.get_prompt() == c["prompt"]
        if c["potential"] is not None:
            assert pb.get_potential_prompt("And then?") == c["potential"]


def test_header_style_prompt_builders_match_reference_strings():
    import merv_amd.prompting as P
    cases = json.loads((G / "prompts_header.json").read_text())  # reference LLaMa31PromptBuilder / Qwen2PromptBuilder
    assert {c["builder"] for c in cases} == {"LLaMa31PromptBuilder", "Qwen2PromptBuilder"}
    for c in cases:
        pb = getattr(P, c["builder"])("merv")
        for t in c["turns"]:
            assert pb.add_turn(t["role"], t["message"]) == t["wrapped"]
        assert pb.get_prompt() == c["prompt"]
    with pytest.raises(TypeError):
        P.Qwen2PromptBuilder("merv", system_prompt="x")


def test_load_video_predecoded_and_gif(tmp_path):
    import numpy as np
    from PIL import Image
    from merv_amd.video_io import load_video
    cases = json.loads((G / "frame_indices.json").read_text())
    c = next(c for c in cases if c["N"] == 300 and c["num_frames"] == 16)
    frames = torch.arange(300, dtype=torch.uint8)[:, None, None, None].expand(300, 4, 6, 3).contiguous()  # frame i is filled with i % 256
    out = load_video((frames, c["fps"]), clip_start_sec=c["clip_start_sec"], clip_end_sec=c["clip_end_sec"], num_frames=16)
    assert out.shape == (16, 3, 4, 6) and out.dtype == torch.uint8
    assert out[:, 0, 0, 0].tolist() == [i % 256 for i in c["ids"]]
    imgs = [Image.fromarray(np.full((8, 10, 3), 20 * i, dtype=np.uint8)) for i in range(10)]
    p = tmp_path / "clip.gif"
    imgs[0].save(p, save_all=True, append_images=imgs[1:], duration=40, loop=0)
    g = load_video(str(p), num_frames=4)
    assert g.shape == (4, 3, 8, 10)
    with pytest.raises(ImportError, match="decord"):
        load_video(str(tmp_path / "missing.mp4"), num_frames=4)


def test_registry_and_model_config_resolution(tmp_path):
    from merv_amd.load import _encoder_weights, available_model_names, available_models, get_model_description
    from merv_amd.registry import GLOBAL_REGISTRY, MODEL_CONFIGS, resolve_model_config
    assert available_models() == ["merv-frozen", "merv-full", "languagebind-single", "dinov2-single", "vivit-single",
                                  "siglip-single"]  # merv/models/registry.py:9-89
    assert GLOBAL_REGISTRY["MERV Full"]["model_id"] == "merv-full" and len(available_model_names()) == 12
    assert get_model_description("merv-full")["optimization_procedure"] == "multi-stage"
    with pytest.raises(ValueError):
        get_model_description("nope")
    full = resolve_model_config({"model_id": "merv-full", "arch_specifier": "3davg+linear", "type": "x", "vidlm_id": "y"})
    assert full["num_frames"] == [16, 16, 32, 16] and full["visual_feature_length"] == 1024 and "type" not in full
    assert full["video_backbone_ids"][0] == "languagebind-video-noclass" and full["feature_fusion"] == "cross_attention_avg_lq"
    assert MODEL_CONFIGS["merv-frozen"]["arch_specifier"] == "no-align+3davg+linear"  # conf/models.py:103 vs :154
    one = resolve_model_config({"model_id": "vivit-single", "video_backbone_ids": ["vivit-google-b-all-no-cls-16frames"],
                                "num_frames": 32})
    assert one["num_frames"] == [32]  # int inflated per backbone, conf/models.py:92-96
    # encoder files: upstream-layout state dicts by backbone id; a missing one is an error
    torch.save({"state_dict": {"a": torch.ones(2)}}, tmp_path / "enc-a.pt")
    got = _encoder_weights(["enc-a"], tmp_path, tmp_path)
    assert torch.equal(got[0]["a"], torch.ones(2))
    with pytest.raises(FileNotFoundError):
        _encoder_weights(["enc-a", "enc-b"], tmp_path, tmp_path)
    assert _encoder_weights(["enc-b"], {"enc-b": "random"}, tmp_path) == ["random"]


def test_llm_registry_families_and_prompt_builders():
    from merv_amd.llm import LLM_BACKBONES, get_llm_backbone_and_tokenizer
    tiny = dict(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                num_key_value_heads=2, max_position_embeddings=128, bos_token_id=1, eos_token_id=2, pad_token_id=0)
    names = {"llama2-7b-pure": "PurePromptBuilder", "llama2-13b-chat": "LLaMa2ChatPromptBuilder",
             "vicuna-v15-7b": "VicunaV15ChatPromptBuilder", "llama3-8b-pure": "PurePromptBuilder",
             "llama3-8b-chat": "LLaMa2ChatPromptBuilder",  # llama3.py:54-55 keeps the Llama-2 chat wrapper
             "llama3.1-8b-chat": "LLaMa31PromptBuilder", "qwen2.5-7b-instruct": "Qwen2PromptBuilder",
             "qwen2.5-3b-instruct": "Qwen2PromptBuilder", "mistral-v0.2-7b-instruct": "MistralInstructPromptBuilder"}
    # every key of the reference's LLM registry (materialize.py:76-101) is present
    for k in ("llama2-7b-pure", "llama2-13b-pure", "llama2-7b-chat", "llama2-13b-chat", "vicuna-v15-7b", "vicuna-v15-13b",
              "llama3-8b-pure", "llama3-8b-chat", "llama3.1-8b-chat", "qwen2.5-7b-instruct", "qwen2.5-3b-instruct"):
        assert k in LLM_BACKBONES
    assert LLM_BACKBONES["llama3.1-8b-chat"][1]()["rope_scaling"]["rope_type"] == "llama3"
    assert LLM_BACKBONES["qwen2.5-3b-instruct"][1]()["tie_word_embeddings"] is True
    for llm_id, builder in names.items():
        assert llm_id in LLM_BACKBONES
        llm, tok = get_llm_backbone_and_tokenizer(llm_id, config=dict(tiny), device="cpu")
        assert tok is None and llm.prompt_builder_fn.__name__ == builder
        ids = llm.generate_from_embeds(torch.randn(1, 4, 32), max_new_tokens=2)
        assert ids.shape == (1, 2) or ids.shape == (1, 1)
    sd = llm.state_dict()
    assert all(k.startswith("llm.") for k in sd)  # the checkpoint's `llm_backbone` key layout (merv.py:282)
    llm2, _ = get_llm_backbone_and_tokenizer("mistral-v0.2-7b-instruct", config=dict(tiny), state_dict=sd, device="cpu", seed=9)
    assert torch.equal(llm2.llm.lm_head.weight, llm.llm.lm_head.weight)
    assert LLM_BACKBONES["mistral-v0.2-7b-instruct"][1]()["num_key_value_heads"] == 8
    with pytest.raises(ValueError):
        get_llm_backbone_and_tokenizer("gpt-17")


def test_training_schedule_and_optimizer_groups():
    from transformers import get_cosine_schedule_with_warmup
    from oracle import merv_oracle as O
    from merv_amd.train import build_optimizer, cosine_with_warmup
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sch = get_cosine_schedule_with_warmup(opt, 3, 40)  # the scheduler the reference builds, fsdp.py:291
    for s in range(45):
        want = opt.param_groups[0]["lr"]
        assert abs(cosine_with_warmup(s, 3, 40) - want) < 1e-12 and abs(O.cosine_schedule_with_warmup(s, 3, 40) - want) < 1e-12
        opt.step()
        sch.step()
    lin = torch.nn.Linear(4, 3)
    ln = torch.nn.LayerNorm(3)
    named = [("projectors.0.projector.projector.weight", lin.weight), ("projectors.0.projector.projector.bias", lin.bias),
             ("llm.norm.weight", ln.weight), ("feature_fusion.Q", torch.nn.Parameter(torch.zeros(1, 8)))]
    o = build_optimizer(named, 1e-3, 0.1)
    decay, no_decay = o.param_groups
    assert decay["weight_decay"] == 0.1 and no_decay["weight_decay"] == 0.0
    assert [p.shape for p in decay["params"]] == [lin.weight.shape, (1, 8)]  # ndim <= 1 or *.bias are not decayed (fsdp.py:281-286)
    assert [p.shape for p in no_decay["params"]] == [lin.bias.shape, ln.weight.shape]


def test_oracle_training_batch_assembly_layout():
    from oracle import merv_oracle as O
    B, S, Cc, Tv = 4, 6, 8, 5
    emb = torch.arange(B * S * Cc, dtype=torch.float32).reshape(B, S, Cc)
    fused = -torch.ones(2, Tv, Cc)
    am = torch.ones(B, S, dtype=torch.bool)
    am[3, 4:] = False
    lab = torch.arange(B * S).reshape(B, S)
    mm = torch.tensor([0, 2])
    e, a, l = O.assemble_training_batch(emb, fused, am, lab, mm, 1)
    assert e.shape == (4, S + Tv, Cc) and a.shape == l.shape == (4, S + Tv)
    # multimodal rows first (in multimodal_indices order): [BOS | visual | rest]
    assert torch.equal(e[1, 0], emb[2, 0]) and torch.equal(e[1, 1:1 + Tv], fused[1]) and torch.equal(e[1, 1 + Tv:], emb[2, 1:])
    assert bool(a[0, 1:1 + Tv].all()) and bool((l[0, 1:1 + Tv] == -100).all()) and l[0, 0] == lab[0, 0]
    # unimodal rows below, padded at the END (merv.py:686-713)
    assert torch.equal(e[2, :S], emb[1]) and not e[2, S:].any() and not a[2, S:].any() and bool((l[2, S:] == -100).all())
    assert torch.equal(a[3, :S], am[3]) and torch.equal(l[3, :S], lab[3])
    logits = torch.randn(4, S + Tv, 11)
    labels = torch.randint(0, 11, (4, S + Tv))
    labels[:, :3] = -100
    want = torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, 11), labels[:, 1:].reshape(-1), ignore_index=-100)
    assert torch.allclose(O.causal_lm_loss(logits, labels), want)


def test_static_decoder_matches_hf_forward_cpu():
    """StaticDecoder (torch-native prefill + static-cache decode, the body of the hipGraph-replayed decode step) against
    the HF module's own forward with its dynamic cache, fp32, Llama (MHA) and Mistral (GQA)."""
    from merv_amd.llm import LlamaBackbone, StaticDecoder
    for fam, kv in (("llama", 4), ("mistral", 2), ("llama3.1", 2), ("qwen2", 2), ("qwen2-tied", 1)):
        cfg = dict(vocab_size=97, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                   num_key_value_heads=kv, max_position_embeddings=64, rms_norm_eps=1e-5, bos_token_id=1, eos_token_id=2, pad_token_id=0)
        if fam == "mistral":
            cfg["sliding_window"] = None
        if fam == "llama3.1":  # frequency-dependent rotary scaling: the tables come from the module's own rotary embedding
            cfg.update(rope_theta=500000.0, max_position_embeddings=256,
                       rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                     "original_max_position_embeddings": 16})
            fam = "llama"
        if fam.startswith("qwen2"):  # q / k / v biases; the 3B model ties lm_head to the embedding
            cfg.update(rope_theta=1e6, tie_word_embeddings=fam.endswith("tied"), use_sliding_window=False)
            fam = "qwen2"
        bb = LlamaBackbone(cfg, device="cpu", dtype=torch.float32, family=fam)
        emb = torch.randn(2, 9, 64, generator=torch.Generator().manual_seed(1))
        dec = StaticDecoder(bb.llm, 16, 2)
        with torch.inference_mode():
            out = bb.llm(inputs_embeds=emb, use_cache=True)
            assert (dec.prefill(emb) - out.logits[:, -1]).abs().max() < 1e-5
            for _ in range(3):
                tok = out.logits[:, -1].argmax(-1)
                out = bb.llm(input_ids=tok[:, None], past_key_values=out.past_key_values, use_cache=True)
                assert (dec.decode(tok, use_graph=False) - out.logits[:, -1]).abs().max() < 1e-5
        # and generate_from_embeds cuts at the step where every row has produced EOS
        ids = bb.generate_from_embeds(emb, max_new_tokens=12, eos_token_id=int(out.logits[0, -1].argmax()), use_graph=False)
        assert 1 <= ids.shape[1] <= 12


def test_load_video_frame_directories(tmp_path):
    """datasets.py:59-112: VLEP directories are 3 fps *.jpg with the clip window, ShareGPT directories *.jpeg sampled over
    the whole clip; anything else raises NotImplementedError, as the reference does."""
    import numpy as np
    from PIL import Image
    from merv_amd.registry import resolve_model_config
    from merv_amd.sampler import frame_indices
    from merv_amd.video_io import load_video
    vlep = tmp_path / "VLEP_clip_001"
    vlep.mkdir()
    for i in range(12):
        Image.fromarray(np.full((6, 8, 3), 20 * i, dtype=np.uint8)).save(vlep / f"{i:05d}.jpg", quality=100)
    out = load_video(str(vlep), clip_start_sec=1.0, clip_end_sec=3.0, num_frames=4)
    want = frame_indices(12, 3.0, 1.0, 3.0, 4, None)  # linspace(3, min(11, 8), 4, dtype=int)
    assert out.shape == (4, 3, 6, 8) and list(want) == [3, 4, 6, 8]
    assert [int(round(float(v) / 20)) for v in out[:, 0, 0, 0]] == list(want)
    sg = tmp_path / "sharegpt_x"
    sg.mkdir()
    for i in range(5):
        Image.fromarray(np.full((4, 4, 3), 40 * i, dtype=np.uint8)).save(sg / f"{i}.jpeg", quality=100)
    out = load_video(str(sg), num_frames=3)
    assert [int(round(float(v) / 40)) for v in out[:, 0, 0, 0]] == [0, 2, 4]
    other = tmp_path / "frames"
    other.mkdir()
    with pytest.raises(NotImplementedError):
        load_video(str(other), num_frames=2)
    cfg = resolve_model_config({"model_id": "merv-full", "num_frames": [16, 16, 32, 16]})
    assert cfg.num_frames == cfg["num_frames"] == [16, 16, 32, 16] and cfg.model_id == "merv-full"
    with pytest.raises(AttributeError):
        cfg.no_such_field


def test_sampler_matches_numpy_on_random_videos():
    """The reference's two call sites are `np.linspace(a, b, n, dtype=int)` (datasets.py:131-141); numpy is on every box, so the
    C sampler is compared with it directly on 3000 random (N, fps, clip window | end_frame, n) tuples beside the committed
    goldens: bit-exact int64 indices, including fractional NTSC rates, windows past the end and fewer frames than samples."""
    import math
    from merv_amd.sampler import frame_indices
    rng = np.random.RandomState(20260)
    rates = [23.976023976023978, 24.0, 25.0, 29.97, 29.97002997002997, 30.0, 50.0, 59.94, 59.94005994005994, 60.0, 12.5, 15.0, 7.0]
    for it in range(3000):
        N = int(rng.randint(1, 40000)) if it % 7 else int(rng.randint(1, 40))
        fps = float(rng.choice(rates)) if it % 5 else float(rng.uniform(1.0, 120.0))
        n = int(rng.choice([1, 2, 4, 8, 12, 16, 32, 33, 64]))
        total = N / fps
        mode = it % 3
        if mode == 0:
            s, e, ef = float(rng.uniform(0, total * 0.7)), None, None
            e = float(rng.uniform(s + 0.01, total * 1.2))
            want = np.linspace(s * fps, min(N - 1, e * fps - 1), n, dtype=int)
        elif mode == 1:
            s, e, ef = 0.0, None, None
            want = np.linspace(s * fps, min(N - 1, total * fps - 1), n, dtype=int)
        else:
            s, e, ef = 0.0, None, int(rng.randint(0, N + 100))
            want = np.linspace(0, min(N - 1, ef), n, dtype=int)
        got = frame_indices(N, fps, s, e, n, ef)
        assert got == [int(i) for i in want], (N, fps, s, e, n, ef)


def test_sampler_under_address_and_ub_sanitizers(tmp_path):
    """The host-side C++ of the library (the frame-index sampler) built with -fsanitize=address,undefined by gcc and driven
    over every committed golden case plus the error paths: no report, same indices. (GPU sanitizers are not available on
    the pool; this is the CPU build the task statement allows.)"""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    cases = json.loads((G / "frame_indices.json").read_text())
    rows = []
    for c in cases:
        s = "NAN" if c["clip_start_sec"] is None or (isinstance(c["clip_start_sec"], float) and c["clip_start_sec"] != c["clip_start_sec"]) else repr(float(c["clip_start_sec"]))
        e = "NAN" if c["clip_end_sec"] is None or (isinstance(c["clip_end_sec"], float) and c["clip_end_sec"] != c["clip_end_sec"]) else repr(float(c["clip_end_sec"]))
        ef = -1 if c["end_frame"] is None else int(c["end_frame"])
        rows.append("{%dLL, %r, %s, %s, %dLL, %d}" % (c["N"], float(c["fps"]), s, e, ef, c["num_frames"]))
    src = tmp_path / "h.cpp"
    src.write_text('''
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include "%s/include/merv_hip.h"
extern "C" void merv_set_error(const char*) {}
struct C { long long N; double fps, s, e; long long ef; int n; };
int main() {
    const C cases[] = {%s};
    for (const C& c : cases) {
        std::vector<int64_t> ids(c.n > 0 ? c.n : 1);
        if (merv_frame_indices(c.N, c.fps, c.s, c.e, c.ef, c.n, ids.data())) return 2;
        for (int i = 0; i < c.n; ++i) printf("%%lld ", (long long)ids[i]);
        printf("\\n");
    }
    int64_t one; int32_t idx[64], cnt;
    if (!merv_frame_indices(0, 30.0, 0.0, NAN, -1, 1, &one)) return 3;      // empty video is an error
    if (!merv_frame_indices(10, 30.0, 0.0, NAN, -1, 1, nullptr)) return 4;  // null output is an error
    if (merv_temporal_subsample(32, 32, 12, idx, &cnt) || cnt != 16) return 5;
    if (!merv_temporal_subsample(32, 16, 32, idx, &cnt)) return 6;          // step 0 is an error
    return 0;
}
''' % (ROOT, ", ".join(rows)))
    exe = tmp_path / "h"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    str(src), str(ROOT / "merv_amd/csrc/sampler.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, env={"ASAN_OPTIONS": "detect_leaks=1"})
    assert out.returncode == 0, out.stderr[-2000:]
    got = [[int(x) for x in line.split()] for line in out.stdout.strip().split("\n")]
    assert got == [c["ids"] for c in cases]


def test_temporal_subsample_matches_the_reference_generate_body():
    """merv_temporal_subsample (host-only C ABI) against the frame selections the reference's own MERV.generate body produced
    (merv.py:796-806; tests/golden/merv_forward.json, tools/make_goldens.py gen_merv_forward)."""
    from merv_amd.sampler import temporal_subsample
    for c in json.loads((G / "merv_forward.json").read_text())["generate_subsample"]:
        for nf, sel in zip(c["num_frames"], c["selected"]):
            assert temporal_subsample(c["frames_loaded"], max(c["num_frames"]), nf) == sel, c


def test_merv_routes_backbones_whose_forward_is_not_the_patch_selection():
    """the visual path drives featurizers directly, so an id that selects class tokens / averages / a pooled head
    must not be fused silently as plain patches (merv.py:563-585 calls vb.forward and reshapes by vb's own resolutions). Since round 5
    such an id goes through its own forward() and the grid the reference derives from it (tests/test_vidlm_gpu.py); what the
    reference's own projector rejects (257 tokens per frame: not H x W) is rejected here too."""
    from merv_amd import backbones as BB
    from merv_amd.vidlm import MERVVisual

    def bare(cls, **attrs):
        o = object.__new__(cls)
        torch.nn.Module.__init__(o)
        for k, v in attrs.items():
            setattr(o, k, v)
        return o
    sel = {
        "languagebind-video-noclass": bare(BB.LangBindVideoBackbone, token="noclass").selects_spec_patches,
        "languagebind-video-classemb": bare(BB.LangBindVideoBackbone, token="classemb").selects_spec_patches,
        "languagebind-video": bare(BB.LangBindVideoBackbone, token=None).selects_spec_patches,
        "dinov2-video-all-tokens": bare(BB.DinoV2VideoBackbone, identifier="dinov2-video-all-tokens").selects_spec_patches,
        "dinov2-video": bare(BB.DinoV2VideoBackbone, identifier="dinov2-video").selects_spec_patches,
        "dinov2-video-all-token-with-cls": bare(BB.DinoV2VideoBackbone, identifier="dinov2-video-all-token-with-cls").selects_spec_patches,
        "vivit-google-b-all-no-cls-16frames": bare(BB.ViVITVideoBackbone, video_backbone_id="vivit-google-b-all-no-cls-16frames").selects_spec_patches,
        "vivit-google-b-all-no-cls": bare(BB.ViVITVideoBackbone, video_backbone_id="vivit-google-b-all-no-cls").selects_spec_patches,
        "vivit-google-b-cls-token": bare(BB.ViVITVideoBackbone, video_backbone_id="vivit-google-b-cls-token").selects_spec_patches,
        "siglip-vit-b16-224px-all-no-cls": bare(BB.SiglipVideoBackbone, class_token=False).selects_spec_patches,
        "siglip-vit-b16-224px": bare(BB.SiglipVideoBackbone, class_token=True).selects_spec_patches,
    }
    assert [k for k, v in sel.items() if v] == ["languagebind-video-noclass", "dinov2-video-all-tokens",
                                                "vivit-google-b-all-no-cls-16frames", "siglip-vit-b16-224px-all-no-cls"]
    vb = bare(BB.LangBindVideoBackbone, token=None, identifier="languagebind-video", num_frames=16)
    assert (vb.num_patches, vb.spatial_resolution, vb.temporal_resolution) == (16 * 257, 257, 16)
    with pytest.raises(ValueError, match="do not form an H x W grid"):
        MERVVisual([vb])
    cls = bare(BB.LangBindVideoBackbone, token="classemb", identifier="languagebind-video-classemb", num_frames=16)
    assert (cls.num_patches, cls.spatial_resolution, cls.temporal_resolution) == (16, 1, 16)  # one token per frame: a 1 x 1 grid, pooled up to 8 x 8


def test_bos_token_length_is_decided_by_the_tokenizer_like_the_reference():
    """merv.py:520-521. Qwen2.5: config.bos_token_id = 151643 but tokenizer.bos_token is None -> 0."""
    from types import SimpleNamespace as NS
    from merv_amd.llm import HFTokenizerAdapter, qwen25_7b_config
    from merv_amd.vidlm import bos_token_length
    assert qwen25_7b_config()["bos_token_id"] is not None  # the trap
    qwen = NS(config=NS(bos_token_id=151643), prepends_bos=False)
    llama = NS(config=NS(bos_token_id=1), prepends_bos=True)
    assert bos_token_length(qwen) == 0 and bos_token_length(llama) == 1
    assert bos_token_length(llama, HFTokenizerAdapter(NS(bos_token=None))) == 0      # the tokenizer wins over the family flag
    assert bos_token_length(qwen, HFTokenizerAdapter(NS(bos_token="<s>"))) == 1
    assert bos_token_length(llama, NS(bos_token=None)) == 0                          # a bare HF tokenizer
    assert bos_token_length(qwen, lambda text: [1, 2]) == 0                          # callable without the attribute -> family flag


def test_distributed_path_argument_forms_are_validated():
    from merv_amd.distributed import DistributedVisualPath
    for kw in (dict(n_videos=2), dict(videos_per_rank=2, n_videos=2), dict(), dict(videos_per_rank=2, n_videos=1, replicate_fusion=True, exchange="all_gather"),
               dict(replicate_fusion=True, exchange="all_gather")):
        with pytest.raises(ValueError, match="form"):
            DistributedVisualPath(None, [], 2, 0, **kw)


def test_bench_starts_its_own_ranks_without_an_outer_launcher():
    """`python3 bench.py --gpus N` (how the driver starts BENCH) must not end in a usage error at N > 1: with WORLD_SIZE unset it
    starts the N ranks itself as a child process, before any GPU call. Here (no GPU) both ranks must get as far as bench.py's own
    "needs a ROCm GPU" exit -- i.e. the launcher ran, each rank imported bench.py with RANK / WORLD_SIZE set -- and the return code
    is the child's."""
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert "starting 2 rank(s) as a child process" in p.stderr, p.stderr[-2000:]
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node=2" in p.stderr
    if not torch.cuda.is_available():
        assert p.returncode != 0
        assert p.stderr.count("bench.py needs a ROCm GPU") >= 1, p.stderr[-2000:]
        assert p.stdout.strip() == ""  # no JSON line without a GPU, and nothing else on stdout either
    # under an outer launcher (WORLD_SIZE set) nothing is spawned
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p2 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env2,
                        capture_output=True, text=True, timeout=600)
    assert "starting" not in p2.stderr


def test_library_override_is_checked_as_strictly_as_the_in_tree_build(monkeypatch, tmp_path):
    """MERV_HIP_LIB alone is a path override: a stale build (missing ABI symbols / another ABI version) must not load. Only
    MERV_HIP_LIB_AB=1 beside it (A/B pairs against a previous round's build) binds what is there and reports the rest."""
    import subprocess
    from merv_amd import _lib
    src = tmp_path / "stale.c"
    src.write_text("int merv_abi_version(void) { return 2; }\nconst char* merv_last_error(void) { return \"\"; }\n")
    so = tmp_path / "libstale.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("MERV_HIP_LIB", str(so))
    monkeypatch.delenv("MERV_HIP_LIB_AB", raising=False)
    with pytest.raises((AttributeError, RuntimeError)):
        _lib.load()
    monkeypatch.setenv("MERV_HIP_LIB_AB", "1")
    lib = _lib.load()
    assert lib.merv_abi_version() == 2
    monkeypatch.setattr(_lib, "_lib", None)


def test_product_library_has_no_environment_hooks(hooks_library):
    """no hidden global state. The product library imports no `getenv` and reports merv_tuning_hooks() == 0; the
    hooks build of the same sources (tests / probes, MERV_TUNING_HOOKS=1 only) exports the same ABI and reports 1. The Python side reads
    its tuning variables through _lib.tuning(), which answers with the default unless MERV_TUNING_HOOKS=1."""
    import os
    import subprocess
    from merv_amd import _lib
    assert hooks_library.merv_tuning_hooks() == 1 and hooks_library.merv_abi_version() == _lib.ABI_VERSION
    for n in _declared_symbols():
        assert hasattr(hooks_library, n), n
    product = C.CDLL(str(_lib.LIB_PATH))
    assert product.merv_tuning_hooks() == 0
    undefined = subprocess.run(["nm", "-D", "--undefined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined, [l for l in undefined.splitlines() if "getenv" in l]
    assert "getenv" in subprocess.run(["nm", "-D", "--undefined-only", str(_lib.HOOKS_LIB_PATH)], capture_output=True, text=True, check=True).stdout
    for src in (ROOT / "merv_amd").glob("*.py"):  # only _lib.py touches the environment for tuning names
        if src.name in ("_lib.py", "load.py"):
            continue
        assert "os.environ" not in src.read_text(), src
    os.environ["MERV_ENCODER_STREAM_MAP"] = "0000"
    try:
        assert _lib.tuning("MERV_ENCODER_STREAM_MAP") == "0000"  # (this test runs with the hooks switch on)
        os.environ["MERV_TUNING_HOOKS"] = "0"
        assert _lib.tuning("MERV_ENCODER_STREAM_MAP") is None and _lib.tuning("MERV_DECODE_GREEDY_GRAPH", "1") == "1"
    finally:
        os.environ.pop("MERV_ENCODER_STREAM_MAP")
        os.environ["MERV_TUNING_HOOKS"] = "1"


@pytest.mark.parametrize("bridge", ["native", "torch"])
def test_load_video_decord_branch_with_a_stand_in_reader(monkeypatch, tmp_path, bridge):
    """merv_amd/video_io.py's .mp4 / .avi branch (datasets.py:125-157: VideoReader(path, ctx=cpu(0)) -> len / get_avg_fps -> the clip's
    frame ids -> get_batch(ids) -> [T,H,W,C] -> [T,C,H,W]). decord is absent from this image, so a stand-in module with the four calls
    the branch makes is injected: get_batch returns an NDArray-like object with .asnumpy() ("native") or a torch tensor (what the
    reference sees after decord.bridge.set_bridge("torch")). The selected frames must be the ones numpy's linspace expression names."""
    import sys
    import types
    from merv_amd.video_io import load_video
    N, fps = 300, 29.97
    clip = np.random.default_rng(0).integers(0, 256, (N, 6, 8, 3), dtype=np.uint8)
    clip[:, 0, 0, 0] = np.arange(N) % 256  # every frame carries its index
    clip[:, 0, 0, 1] = np.arange(N) // 256
    calls = {}

    class _ND:
        def __init__(self, a):
            self._a = a

        def asnumpy(self):
            return self._a

    class VideoReader:
        def __init__(self, path, ctx=None):
            calls["path"], calls["ctx"] = path, ctx

        def __len__(self):
            return N

        def get_avg_fps(self):
            return fps

        def get_batch(self, ids):
            calls["ids"] = [int(i) for i in ids]
            a = clip[np.asarray(calls["ids"])]
            return torch.from_numpy(a) if bridge == "torch" else _ND(a)

    mod = types.ModuleType("decord")
    mod.VideoReader, mod.cpu = VideoReader, (lambda i=0: ("cpu", i))
    monkeypatch.setitem(sys.modules, "decord", mod)
    path = tmp_path / "clip.mp4"
    path.write_bytes(b"")
    for kw in (dict(num_frames=32), dict(num_frames=16, clip_start_sec=2.0, clip_end_sec=7.5), dict(num_frames=8, end_frame=120),
               dict(num_frames=16, clip_start_sec=float("nan"), clip_end_sec=float("nan"))):
        out = load_video(str(path), **kw)
        n = kw["num_frames"]
        start = 0.0 if kw.get("clip_start_sec") is None or np.isnan(kw.get("clip_start_sec", 0.0)) else kw["clip_start_sec"]
        end = kw.get("clip_end_sec")
        end = N / fps if end is None or np.isnan(end) else end
        if kw.get("end_frame") is not None:
            want = np.linspace(0, min(N - 1, kw["end_frame"]), n, dtype=int)
        else:
            want = np.linspace(start * fps, min(N - 1, end * fps - 1), n, dtype=int)  # datasets.py:131-141
        assert calls["ids"] == want.tolist() and calls["path"] == str(path) and calls["ctx"] == ("cpu", 0)
        assert out.dtype == torch.uint8 and tuple(out.shape) == (n, 3, 6, 8) and out.is_contiguous()
        got = out[:, 0, 0, 0].long() + 256 * out[:, 1, 0, 0].long()
        assert got.tolist() == want.tolist()
        assert torch.equal(out, torch.from_numpy(clip[want]).permute(0, 3, 1, 2))
    # without decord: a loud ImportError, never a silent fallback
    monkeypatch.setitem(sys.modules, "decord", None)
    with pytest.raises(ImportError, match="decord"):
        load_video(str(path), num_frames=8)
