"""GPU: `load_vid()` on a run directory laid out like the reference's (config.json + checkpoints/latest-checkpoint.pt,
load_vid.py:46-127), BASELINE.json configs[0]: the DINOv2-only single-encoder registry model on 4 frames with greedy
decode. Encoder parameters are read from a timm-layout state-dict file; the LLM is a reduced-geometry Llama."""
import json

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TINY_LLM = dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=4, max_position_embeddings=2048, rms_norm_eps=1e-5, bos_token_id=1, eos_token_id=2,
                pad_token_id=0)


def _write_run(tmp_path, dev):
    from merv_amd.backbones import random_weights
    from merv_amd.encoder import EncoderSpec
    from merv_amd.llm import LlamaBackbone
    from merv_amd.projector import AveragePooling3DProjector
    from merv_amd.weights import to_timm_names
    run = tmp_path / "dinov2-single+4f"
    (run / "checkpoints").mkdir(parents=True)
    (run / "encoders").mkdir()
    cfg = {"model": {"type": "dinov2-single", "vidlm_id": None, "model_id": "dinov2-single",
                     "arch_specifier": "no-align+3davg+linear", "feature_fusion": None,
                     "video_backbone_ids": ["dinov2-video-all-tokens"], "llm_backbone_id": "llama2-7b-pure",
                     "image_resize_strategy": "resize-naive", "llm_max_length": 2048, "num_frames": 4,
                     "projector_token_length": 64, "visual_feature_length": 1024}}
    cfg["model"].pop("vidlm_id")
    (run / "config.json").write_text(json.dumps(cfg))
    spec = EncoderSpec("dinov2", 1024, 16, 4096, 23, 14, 1, 224, 4, "BFCHW", 5, False, False, False, True, 0, "gelu_erf", 1e-6)
    canon = random_weights(spec, seed=7)
    torch.save(to_timm_names(canon), run / "encoders" / "dinov2-video-all-tokens.pt")
    torch.manual_seed(3)
    proj = AveragePooling3DProjector(1024, 256, output_frames=4, output_size=8, mlp_type="linear")
    llm = LlamaBackbone(TINY_LLM, device="cpu", seed=5)
    ckpt = {"model": {"projector": {k: v.clone() for k, v in proj.state_dict().items()},  # legacy single-projector key, merv.py:273
                      "llm_backbone": {k: v.clone() for k, v in llm.state_dict().items()}}}
    torch.save(ckpt, run / "checkpoints" / "latest-checkpoint.pt")
    return run, spec, canon, proj, llm


def test_load_vid_config0_dinov2_single_greedy(tmp_path, dev):
    from oracle import merv_oracle as O
    from merv_amd.load import available_models, load_vid
    assert "dinov2-single" in available_models()
    run, spec, canon, proj, llm_cpu = _write_run(tmp_path, dev)
    vidlm, cfg = load_vid(run, get_model_cfg=True, llm_config=TINY_LLM, device=dev)
    assert cfg["num_frames"] == [4] and cfg["model_id"] == "dinov2-single"
    assert vidlm.visual_feature_length == 256 and vidlm.feature_fusion is None
    assert type(vidlm.get_prompt_builder()).__name__ == "PurePromptBuilder"
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (4, 3, 240, 320), generator=g, dtype=torch.uint8)
    prompt_ids = [1, 17, 44, 9, 201]
    ids = vidlm.generate((frames, 30.0), prompt_ids, [4], max_new_tokens=6)
    assert ids.shape[0] == 1 and 1 <= ids.shape[1] <= 6
    # (1) the loaded visual branch against the oracle on the same pixels and parameters
    pix = vidlm.video_backbones[0].video_transform(frames.to(dev)).unsqueeze(0)
    fused, w = vidlm.encode([pix])
    assert w is None and fused.shape == (1, 256, 256)
    ocfg = O.EncoderCfg(**{k: getattr(spec, k) for k in O.EncoderCfg.__dataclass_fields__})
    tok = O.encoder_forward(pix.float().cpu(), ocfg, canon)
    lin = proj.projector.projector
    ref = O.projector_forward(tok, 4, 16, 8, lin.weight.detach(), lin.bias.detach())
    assert rel_l2(fused, ref) < 2e-2
    # (2) the decode loop against a plain greedy loop on the CPU copy of the same LLM over the same embeddings
    emb = vidlm.llm_backbone.embed_input_ids(torch.tensor([prompt_ids], device=dev))
    full, _, _, _ = vidlm.forward_visual([pix], emb)
    x = full.float().cpu()
    ref_llm = llm_cpu.llm.float()
    emb_w = ref_llm.get_input_embeddings().weight
    greedy = []
    for _ in range(ids.shape[1]):
        nxt = int(ref_llm(inputs_embeds=x).logits[0, -1].argmax())
        greedy.append(nxt)
        x = torch.cat([x, emb_w[nxt][None, None]], 1)
    assert greedy[0] == int(ids[0, 0])  # later tokens may legitimately flip on bf16 near-ties
    agree = sum(int(a == b) for a, b in zip(greedy, ids[0].tolist()))
    assert agree >= ids.shape[1] - 2, (greedy, ids)


def test_load_vid_errors(tmp_path, dev):
    from merv_amd.load import load_vid
    with pytest.raises(ValueError):
        load_vid("no-such-model")
    with pytest.raises(FileNotFoundError):
        load_vid("merv-full", cache_dir=tmp_path)  # registry id, nothing on disk, no hub
    run, *_ = _write_run(tmp_path, dev)
    (run / "encoders" / "dinov2-video-all-tokens.pt").unlink()
    with pytest.raises(FileNotFoundError):
        load_vid(run, llm_config=TINY_LLM, device=dev)  # never a silent random encoder


def test_quick_start_script_shape(tmp_path, dev):
    """The reference's scripts/quick_start.py, line for line, against a local run directory: `from merv import load_vid`,
    `vidlm.to(device, dtype=bf16)`, prompt builder turns, `generate(video_path, prompt_text, num_frames=..., do_sample=...,
    temperature=..., max_new_tokens=..., min_length=...)`. The video is a GIF (no decord here), the tokenizer a stand-in."""
    import numpy as np
    from PIL import Image
    from merv import load_vid  # the alias package at the repo root

    class Tok:  # callable tokenizer with decode(), the two things generate() uses
        def __call__(self, text):
            return [1] + [3 + (ord(c) % 200) for c in text][:20]

        def decode(self, ids):
            return " ".join(str(i) for i in ids)

    run, *_ = _write_run(tmp_path, dev)
    vidlm = load_vid(str(run), hf_token="unused", llm_config=TINY_LLM, tokenizer=Tok(), device=dev)
    vidlm.to(dev, dtype=torch.bfloat16)
    frames = [Image.fromarray(np.full((48, 64, 3), 25 * i, dtype=np.uint8)) for i in range(8)]
    video_path = tmp_path / "clip.gif"
    frames[0].save(video_path, save_all=True, append_images=frames[1:], duration=40, loop=0)
    prompt_builder = vidlm.get_prompt_builder()
    prompt_builder.add_turn(role="human", message="Describe what is happening in this video.")
    prompt_text = prompt_builder.get_prompt()
    assert prompt_text == "In: Describe what is happening in this video.\nOut:"
    generated_text = vidlm.generate(str(video_path), prompt_text, num_frames=[4], do_sample=True, temperature=0.4,
                                    max_new_tokens=5, min_length=1)
    assert isinstance(generated_text, str) and 1 <= len(generated_text.split()) <= 5


def test_eval_mcq_script_shape(tmp_path, dev, monkeypatch):
    """The call sequence of the reference's scripts/eval_mcq.py:100-160, in its order, through the `merv` alias:
    `from merv.models.load_vid import load_vid`, `load_vid(run, hf_token=..., get_model_cfg=True)`, `.to(device, bf16)`, per question
    `vidlm.llm_backbone.prompt_builder_fn(model_family="merv")`, `add_turn`, `get_prompt`, then `generate(video_name, prompt_text,
    do_sample=..., temperature=..., max_new_tokens=..., min_length=..., num_frames=model_cfg.num_frames, clip_start_sec=...,
    clip_end_sec=..., end_frame=...)` with the dummy_mcq question's `end_frame = 595`. The clip is pre-decoded (decord is absent): a
    (frames, fps) pair of 900 frames stands for the file, and the frames the library selects are checked against the reference's
    expression (datasets.py:131-141: np.linspace(start, min(end, end_frame), n, dtype=int)) and against what reaches the encoder."""
    import numpy as np
    from merv.models.load_vid import load_vid  # the alias package at the repo root, as the script imports it
    import merv_amd.video_io as vio

    class Tok:
        def __call__(self, text):
            self.last_text = text
            return [1] + [3 + (ord(c) % 200) for c in text][:24]

        def decode(self, ids):
            return " ".join(str(i) for i in ids)

    run, *_ = _write_run(tmp_path, dev)
    tok = Tok()
    vidlm, model_cfg = load_vid(str(run), hf_token="unused", get_model_cfg=True, llm_config=TINY_LLM, tokenizer=tok, device=dev)
    vidlm.to(dev, dtype=torch.bfloat16)
    num_frames = model_cfg["num_frames"] if isinstance(model_cfg, dict) else model_cfg.num_frames
    assert list(num_frames) == [4]

    # frame i is filled with the value i % 251 (and i // 251 in the blue channel), so a selected frame names its index
    n_frames, fps = 900, 29.97
    idx = torch.arange(n_frames)
    clip = torch.zeros(n_frames, 24, 32, 3, dtype=torch.uint8)
    clip[..., 0] = (idx % 251)[:, None, None].to(torch.uint8)
    clip[..., 1] = (idx % 251)[:, None, None].to(torch.uint8)
    clip[..., 2] = (idx // 251)[:, None, None].to(torch.uint8)
    seen = {}
    orig = vio.load_video

    def spy(video, **kw):
        out = orig(video, **kw)
        seen["kw"] = kw
        seen["ids"] = (out[:, 0, 0, 0].long() + 251 * out[:, 2, 0, 0].long()).tolist()
        return out

    monkeypatch.setattr(vio, "load_video", spy)
    question = {"question_id": 0, "video_name": "dummy", "end_frame": 595, "question": "What is shown?", "a0": "a cat", "a1": "a dog"}
    prompt_builder = vidlm.llm_backbone.prompt_builder_fn(model_family="merv")
    question_text = question["question"] + " Options: (A) a cat (B) a dog. Answer with the option's letter."
    prompt_builder.add_turn(role="human", message=question_text)
    prompt_text = prompt_builder.get_prompt()
    assert prompt_text == f"In: {question_text}\nOut:"
    clip_start_sec = question["time"][0] if "time" in question else 0.0
    clip_end_sec = question["time"][1] if "time" in question else None
    end_frame = question["end_frame"] if "end_frame" in question else None
    generated_text = vidlm.generate((clip, fps), prompt_text, do_sample=False, temperature=0.0, max_new_tokens=4, min_length=1,
                                    num_frames=num_frames, clip_start_sec=clip_start_sec, clip_end_sec=clip_end_sec, end_frame=end_frame)
    assert isinstance(generated_text, str) and 1 <= len(generated_text.split()) <= 4
    assert tok.last_text == prompt_text
    # datasets.py:131-141 with clip_start_sec = 0, clip_end_sec = None, end_frame = 595: linspace(0, min(N - 1, 595), 4) as ints
    expect = np.linspace(0, min(n_frames - 1, 595), max(num_frames), dtype=int).tolist()
    assert seen["kw"]["end_frame"] == 595 and seen["kw"]["num_frames"] == max(num_frames)
    assert seen["ids"] == expect == [0, 198, 396, 595]


def test_eval_openended_script_shape(tmp_path, dev, monkeypatch):
    """The loop of the reference's scripts/eval_openended.py:56-219, in its order, through the `merv` alias: questions / answers json,
    `get_chunk`, `load_vid(run, hf_token=...)` + `.to(device, bf16)`, the resume file, then per question a fresh
    `vidlm.llm_backbone.prompt_builder_fn(model_family="merv")` turn (with the "\\n<video>" suffix of the `_token` datasets),
    `glob` for the video file by name, `vidlm.generate(video_name, prompt_text, do_sample=, temperature=, max_new_tokens=, min_length=,
    num_frames=)` inside `try / except Exception: continue`, one jsonl line per answered question merged with its answer record, and
    the final rename to `_done.jsonl`. One of the three videos is unreadable: the loop must print, skip it and go on (:189-192), and
    the model must still answer the next question. The time-window form of the same call (`clip_start_sec` / `clip_end_sec`, as
    eval_mcq.py passes them from question["time"]) is checked on a pre-decoded clip against datasets.py:131-136."""
    import glob
    import math
    import os
    import numpy as np
    from PIL import Image
    from merv.models.load_vid import load_vid
    import merv_amd.video_io as vio

    class Tok:
        def __call__(self, text):
            self.last_text = text
            return [1] + [3 + (ord(c) % 200) for c in text][:24]

        def decode(self, ids):
            return " ".join(str(i) for i in ids)

    # ---- eval_data/<benchmark>/{test_q,test_a}.json + videos/, as the script reads them (:62-77)
    benchmark, eval_dataset = "MSVDsmall", "MSVDsmall_token"
    data = tmp_path / "eval_data" / benchmark
    (data / "videos").mkdir(parents=True)
    questions = [{"question_id": f"q{i}", "video_name": f"vid{i}", "question": q} for i, q in
                 enumerate(["what is the man doing?", "who is running?", "what color is the car?"])]
    answers = [{"question_id": f"q{i}", "answer": a, "answer_type": 0} for i, a in enumerate(["cooking", "a dog", "red"])]
    (data / "test_q.json").write_text(json.dumps(questions))
    (data / "test_a.json").write_text(json.dumps(answers))
    for i in (0, 2):
        frames = [Image.fromarray(np.full((48, 64, 3), 20 * k + 5 * i, dtype=np.uint8)) for k in range(9)]
        frames[0].save(data / "videos" / f"vid{i}.gif", save_all=True, append_images=frames[1:], duration=40, loop=0)
    (data / "videos" / "vid1.gif").write_bytes(b"this is not a gif")  # the unreadable video

    def split_list(lst, n):  # eval_openended.py:22-25
        chunk_size = math.ceil(len(lst) / n)
        return [lst[i:i + chunk_size] for i in range(0, len(lst), chunk_size)]

    num_chunks, chunk_idx = 1, 0
    qs = json.load(open(data / "test_q.json"))
    all_questions_id = set(item["question_id"] for item in qs)
    qs = split_list(qs, num_chunks)[chunk_idx]
    answers_dict = {item["question_id"]: item for item in json.load(open(data / "test_a.json"))}

    run, *_ = _write_run(tmp_path, dev)
    tok = Tok()
    vidlm, model_cfg = load_vid(str(run), hf_token="unused", get_model_cfg=True, llm_config=TINY_LLM, tokenizer=tok, device=dev)
    vidlm.to(dev, dtype=torch.bfloat16)
    num_frames = model_cfg["num_frames"] if isinstance(model_cfg, dict) else model_cfg.num_frames

    result_dir = tmp_path / "eval_result" / "run"
    os.makedirs(result_dir, exist_ok=True)
    pred = result_dir / f"{eval_dataset}_pred_{num_chunks}_{chunk_idx}.jsonl"
    done_lines, failed = [], []
    with open(pred, "w") as f:
        for line in done_lines:
            f.write(line)
        for i, question in enumerate(qs):
            prompt_builder = vidlm.llm_backbone.prompt_builder_fn(model_family="merv")
            message = question["question"] + ("\n<video>" if "_token" in eval_dataset else "")
            prompt_builder.add_turn(role="human", message=message)
            prompt_text = prompt_builder.get_prompt()
            video_name = glob.glob(f"{data}/videos/{question['video_name']}.*")[0]
            try:
                generated_text = vidlm.generate(video_name, prompt_text, do_sample=False, temperature=1.0, max_new_tokens=6, min_length=1,
                                                num_frames=num_frames)
                question["pred"] = generated_text
                question["message"] = message
                question = {**question, **answers_dict[question["question_id"]]}
                f.write(json.dumps(question) + "\n")
            except Exception as e:  # if video loading has an issue (:189-192)
                print(e)
                failed.append(video_name)
                continue
    os.rename(pred, result_dir / f"{eval_dataset}_pred_{num_chunks}_{chunk_idx}_done.jsonl")
    lines = [json.loads(l) for l in open(result_dir / f"{eval_dataset}_pred_{num_chunks}_{chunk_idx}_done.jsonl")]
    assert [l["question_id"] for l in lines] == ["q0", "q2"] and len(failed) == 1 and failed[0].endswith("vid1.gif")
    assert all(isinstance(l["pred"], str) and 1 <= len(l["pred"].split()) <= 6 for l in lines)
    assert lines[0]["answer"] == "cooking" and lines[1]["answer"] == "red" and lines[1]["message"].endswith("\n<video>")
    assert tok.last_text == f"In: {questions[2]['question']}\n<video>\nOut:"
    assert all_questions_id - set(l["question_id"] for l in lines) == {"q1"}  # not merged: one question is still open (:211-218)
    # the same question asked again gives the same greedy answer (the failed call in between left the decoder usable)
    again = vidlm.generate(str(data / "videos" / "vid0.gif"), f"In: {questions[0]['question']}\n<video>\nOut:", do_sample=False, temperature=1.0,
                           max_new_tokens=6, min_length=1, num_frames=num_frames)
    assert again == lines[0]["pred"]

    # ---- the time-window form: clip_start_sec / clip_end_sec in seconds select frames [start * fps, end * fps] (datasets.py:131-141)
    n_frames, fps = 600, 25.0
    idx = torch.arange(n_frames)
    clip = torch.zeros(n_frames, 24, 32, 3, dtype=torch.uint8)
    clip[..., 0] = (idx % 251)[:, None, None].to(torch.uint8)
    clip[..., 2] = (idx // 251)[:, None, None].to(torch.uint8)
    seen = {}
    orig = vio.load_video

    def spy(video, **kw):
        out = orig(video, **kw)
        seen["kw"] = kw
        seen["ids"] = (out[:, 0, 0, 0].long() + 251 * out[:, 2, 0, 0].long()).tolist()
        return out

    monkeypatch.setattr(vio, "load_video", spy)
    question = {"time": [4.0, 12.5]}
    text = vidlm.generate((clip, fps), "In: what happens?\nOut:", do_sample=False, temperature=1.0, max_new_tokens=3, min_length=1,
                          num_frames=num_frames, clip_start_sec=question["time"][0], clip_end_sec=question["time"][1])
    assert isinstance(text, str)
    # datasets.py:131-136: np.linspace(clip_start_sec * avg_fps, min(N - 1, clip_end_sec * avg_fps - 1), num_frames, dtype=int)
    expect = np.linspace(4.0 * fps, min(n_frames - 1, 12.5 * fps - 1), max(num_frames), dtype=int).tolist()
    assert seen["kw"]["clip_start_sec"] == 4.0 and seen["kw"]["clip_end_sec"] == 12.5
    assert seen["ids"] == expect, (seen["ids"], expect)
