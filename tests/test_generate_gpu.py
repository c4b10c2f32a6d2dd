"""GPU: the front door end to end -- pre-decoded clip -> frame indices -> per-encoder subsample -> GPU transforms -> HIP
encoders -> projectors -> fusion -> splice -> LLM prefill + greedy decode (tiny random Llama, PyTorch-ROCm)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class ByteTokenizer:
    def __call__(self, text):
        return [1] + [3 + b for b in text.encode("utf-8")]

    def decode(self, ids):
        return bytes([max(0, min(255, i - 3)) for i in ids]).decode("latin-1")


def _model(dev, ids, frames, tokenizer=None, vfl=1024):
    from merv_amd.backbones import VIDEO_BACKBONES
    from merv_amd.llm import LlamaBackbone
    from merv_amd.vidlm import MERV
    bbs = [VIDEO_BACKBONES[i]["cls"](i, "resize-naive", num_frames=f, weights="random", device=dev, layers=1,
                                     **VIDEO_BACKBONES[i]["kwargs"]) for i, f in zip(ids, frames)]
    llm = LlamaBackbone(dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                             num_key_value_heads=4, max_position_embeddings=4096, bos_token_id=1, eos_token_id=2, pad_token_id=0),
                        device=dev)
    return MERV(bbs, llm, tokenizer=tokenizer, visual_feature_length=vfl)


def test_generate_merv_full_shape(dev):
    ids = ["languagebind-video-noclass", "dinov2-video-all-tokens", "vivit-google-b-all-no-cls-16frames",
           "siglip-vit-b16-224px-all-no-cls"]
    m = _model(dev, ids, [16, 16, 32, 16], ByteTokenizer())
    clip = (torch.randint(0, 256, (90, 120, 160, 3), dtype=torch.uint8), 29.97)  # decoded frames [N,H,W,3] + fps
    pb = m.get_prompt_builder()
    pb.add_turn("human", "What is happening?")
    text = m.generate(clip, pb.get_prompt(), [16, 16, 32, 16], max_new_tokens=5)
    assert isinstance(text, str)
    w = m.last_fusion_weights
    assert w.shape == (1, 4) and abs(float(w.sum()) - 1.0) < 1e-4
    out_ids = m.generate(clip, [1, 5, 6, 7], [16, 16, 32, 16], max_new_tokens=5)  # deterministic greedy
    again = m.generate(clip, [1, 5, 6, 7], [16, 16, 32, 16], max_new_tokens=5)
    assert out_ids == again


def test_generate_equals_manual_prefill(dev):
    """generate() == embed -> splice fused tokens after BOS -> HF forward -> argmax, done by hand (config 1: DINOv2 only)."""
    from merv_amd.sampler import temporal_subsample
    from merv_amd.video_io import load_video
    m = _model(dev, ["dinov2-video-all-tokens"], [4])
    m.tokenizer = None
    clip = (torch.randint(0, 256, (40, 3, 64, 80), dtype=torch.uint8), 25.0)
    prompt = [1, 9, 8, 7, 6]
    ids = m.generate(clip, prompt, [4], max_new_tokens=3)
    assert ids.shape == (1, 3)
    frames = load_video(clip, num_frames=4).to(dev)
    pix = m.video_backbones[0].video_transform(frames[temporal_subsample(4, 4, 4)].contiguous())[None]
    fused, _ = m.encode([pix])
    emb = m.llm_backbone.embed_input_ids(torch.tensor([prompt], device=dev))
    full = torch.cat([emb[:, :1], fused.to(emb.dtype), emb[:, 1:]], 1)
    logits = m.llm_backbone.llm(inputs_embeds=full).logits[:, -1]
    assert int(logits.argmax(-1)) == int(ids[0, 0])
    assert fused.shape == (1, 256, 256)


def test_graph_decode_matches_eager_static_decode(dev):
    """generate_from_embeds(use_graph=True): the hipGraph-replayed decode step reproduces the same step run eagerly, token
    for token; and the torch-native static-cache path agrees with the HF forward's logits within bf16 noise."""
    from merv_amd.llm import LlamaBackbone, StaticDecoder
    cfg = dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4,
               num_key_value_heads=2, max_position_embeddings=512, rms_norm_eps=1e-5, bos_token_id=1, eos_token_id=None, pad_token_id=0)
    bb = LlamaBackbone(cfg, device=dev, dtype=torch.bfloat16, family="llama", seed=3)
    g = torch.Generator().manual_seed(0)
    emb = (torch.randn(1, 40, 256, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    ids_graph = bb.generate_from_embeds(emb, max_new_tokens=24, use_graph=True)
    dec = StaticDecoder(bb.llm, 40 + 25, 1)
    logits = dec.prefill(emb)
    eager = []
    for _ in range(24):
        tok = logits.argmax(-1)
        eager.append(int(tok))
        logits = dec.decode(tok, use_graph=False)
    assert ids_graph[0].tolist() == eager
    with torch.inference_mode():
        hf = bb.llm(inputs_embeds=emb).logits[:, -1].float()
    mine = StaticDecoder(bb.llm, 64, 1).prefill(emb)
    assert float((mine - hf).norm() / hf.norm()) < 2e-2


def test_generate_still_image_and_decoding_kwargs(dev, tmp_path):
    """merv.py:787-793: a `.jpg` path is a still image repeated max(num_frames) times, then the per-encoder stride. And the
    kwargs the reference's scripts pass (scripts/quick_start.py:24-32, eval_mcq.py:146-153: do_sample, temperature,
    max_new_tokens, min_length) are honoured; an argument the decode loop does not implement fails loudly."""
    import numpy as np
    from PIL import Image
    m = _model(dev, ["dinov2-video-all-tokens", "siglip-vit-b16-224px-all-no-cls"], [4, 4], vfl=256)
    m.tokenizer = None
    img = (np.random.default_rng(0).integers(0, 256, (96, 128, 3))).astype(np.uint8)
    path = tmp_path / "frame.jpg"
    Image.fromarray(img).save(path, quality=95)
    ids = m.generate(str(path), [1, 9, 8], [4, 4], max_new_tokens=4, do_sample=False, temperature=1.0, min_length=1)
    assert ids.shape == (1, 4)
    # the same thing by hand: decode the jpeg once, repeat, stride, transform
    dec = torch.from_numpy(np.array(Image.open(path).convert("RGB")).transpose(2, 0, 1)[None].repeat(4, 0)).to(dev)
    vv = [vb.video_transform(dec[:: 4 // nf].contiguous())[None] for vb, nf in zip(m.video_backbones, [4, 4])]
    fused, _ = m.encode(vv)
    fused = fused.clone()
    emb = m.llm_backbone.embed_input_ids(torch.tensor([[1, 9, 8]], device=dev))
    full = torch.cat([emb[:, :1], fused.to(emb.dtype), emb[:, 1:]], 1)
    first = int(m.llm_backbone.llm(inputs_embeds=full).logits[:, -1].argmax(-1))
    assert int(ids[0, 0]) == first
    # sampling controls: top_k = 1 is greedy whatever the temperature; a seeded sampler repeats
    assert torch.equal(m.generate(str(path), [1, 9, 8], [4, 4], max_new_tokens=4, do_sample=True, temperature=0.7, top_k=1), ids)
    a = m.generate(str(path), [1, 9, 8], [4, 4], max_new_tokens=6, do_sample=True, temperature=0.9, top_p=0.8, repetition_penalty=1.2)
    assert a.shape[1] <= 6 and int(a.max()) < 320
    # min_length counts the prompt: EOS (id 2) cannot appear among the first tokens
    b = m.generate(str(path), [1, 9, 8], [4, 4], max_new_tokens=5, min_length=3 + 5)
    assert b.shape == (1, 5) and not bool((b == 2).any())
    with pytest.raises(NotImplementedError):
        m.generate(str(path), [1, 9, 8], [4, 4], max_new_tokens=2, num_beams=4)
    with pytest.raises(TypeError, match="unsupported generation arguments"):
        m.generate(str(path), [1, 9, 8], [4, 4], max_new_tokens=2, length_penalty=2.0)


def test_repetition_penalty_covers_the_prompt_like_hf_generate(dev):
    """The reference hands input_ids to HF generate (merv.py:819), so RepetitionPenaltyLogitsProcessor penalises prompt tokens
    as well as generated ones. Greedy decode with a strong penalty: every step's choice must equal the argmax of
    HF's own processor applied to (prompt ids + tokens so far); without the prompt ids the first choice differs. Also: top_p /
    repetition_penalty = None (a caller forwarding HF-style `None`) are accepted."""
    from transformers.generation.logits_process import RepetitionPenaltyLogitsProcessor
    from merv_amd.llm import LlamaBackbone
    llm = LlamaBackbone(dict(vocab_size=64, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                             num_key_value_heads=4, max_position_embeddings=512, bos_token_id=1, eos_token_id=None, pad_token_id=0), device=dev)
    prompt = torch.tensor([[1, 7, 9, 11, 13, 15, 17]], device=dev)
    emb = llm.embed_input_ids(prompt)
    pen = 5.0
    got = llm.generate_from_embeds(emb, max_new_tokens=6, repetition_penalty=pen, prompt_ids=prompt, use_graph=False)
    # replay with HF's processor on the module's own forward
    proc = RepetitionPenaltyLogitsProcessor(penalty=pen)
    ids, cur = prompt.clone(), emb
    for i in range(got.shape[1]):
        logits = llm.llm(inputs_embeds=cur.to(llm.dtype)).logits[:, -1].float()
        nxt = proc(ids, logits).argmax(-1)
        assert int(nxt) == int(got[0, i]), (i, int(nxt), got.tolist())
        ids = torch.cat([ids, nxt[:, None]], 1)
        cur = torch.cat([cur, llm.embed_input_ids(nxt[:, None])], 1)
    # the prompt's own ids are never repeated under this penalty unless nothing else is left
    plain = llm.generate_from_embeds(emb, max_new_tokens=6, repetition_penalty=pen, use_graph=False)
    assert plain.shape == got.shape
    none_ok = llm.generate_from_embeds(emb, max_new_tokens=3, do_sample=True, top_p=None, repetition_penalty=None, top_k=None, use_graph=False,
                                       generator=torch.Generator(device=dev).manual_seed(0))
    assert none_ok.shape == (1, 3)
