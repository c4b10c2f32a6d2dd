"""GPU: the batch-1 decode kernels of the LLM hand-off (csrc/decode.hip, row f-3) against plain torch restatements of the
HF module's ops (transformers modeling_llama: LlamaRMSNorm, nn.Linear, apply_rotary_pos_emb, eager attention, LlamaMLP), and
the decoder built from them (merv_amd.llm.HipDecoder) against the PyTorch-ROCm StaticDecoder on the same random model."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _st(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def test_rmsnorm(dev):
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    for rows, D in [(1, 4096), (3, 256), (1, 5120)]:
        x = (torch.randn(rows, D, generator=g) * 3).to(torch.bfloat16).to(dev)
        w = (1 + 0.2 * torch.randn(D, generator=g)).to(torch.bfloat16).to(dev)
        y = torch.empty_like(x)
        check(lib.merv_decode_rmsnorm(ptr(x), ptr(w), ptr(y), rows, D, 1e-5, _st(dev)), "rms")
        xf = x.float()
        ref = w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(torch.bfloat16)  # LlamaRMSNorm.forward
        assert (y.float() - ref.float()).abs().max() <= 2 * ref.float().abs().max() * 2**-8  # at most one bf16 ulp apart
        assert rel_l2(y, ref) < 2e-3


@pytest.mark.parametrize("N,K", [(4096, 4096), (11008, 4096), (4096, 11008), (1000, 264), (32064, 4096), (3, 8)])
def test_gemv_forms(dev, N, K):
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(N + K)
    W = (torch.randn(N, K, generator=g) * K**-0.5).to(torch.bfloat16).to(dev)
    W2 = (torch.randn(N, K, generator=g) * K**-0.5).to(torch.bfloat16).to(dev)
    x = torch.randn(K, generator=g).to(torch.bfloat16).to(dev)
    res = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
    lin = (W.float() @ x.float())
    y = torch.empty(N, dtype=torch.bfloat16, device=dev)
    check(lib.merv_decode_gemv(ptr(W), 0, ptr(x), 0, ptr(y), 0, N, K, 0, 0.0, _st(dev)), "gemv")
    assert rel_l2(y, lin) < 4e-3
    y32 = torch.empty(N, dtype=torch.float32, device=dev)
    check(lib.merv_decode_gemv(ptr(W), 0, ptr(x), 0, 0, ptr(y32), N, K, 0, 0.0, _st(dev)), "gemv32")
    assert torch.equal(y32, y.float())  # logits = .float() of the bf16 linear output
    r = res.clone()
    check(lib.merv_decode_gemv(ptr(W), 0, ptr(x), ptr(r), ptr(r), 0, N, K, 0, 0.0, _st(dev)), "gemv+res")  # in place: x += linear
    assert rel_l2(r, res.float() + lin.to(torch.bfloat16).float()) < 4e-3
    check(lib.merv_decode_gemv(ptr(W), ptr(W2), ptr(x), 0, ptr(y), 0, N, K, 0, 0.0, _st(dev)), "gated")
    ref = F.silu(lin.to(torch.bfloat16)) * (W2.float() @ x.float()).to(torch.bfloat16)  # act_fn(gate_proj(x)) * up_proj(x)
    assert rel_l2(y, ref) < 8e-3
    with pytest.raises(ValueError):
        check(lib.merv_decode_gemv(ptr(W), ptr(W2), ptr(x), ptr(res), ptr(y), 0, N, K, 0, 0.0, _st(dev)), "gated+res")
    # RMSNorm fused into the launch == merv_decode_rmsnorm followed by the plain launch, bit for bit (plain and gated form)
    wn = (1 + 0.2 * torch.randn(K, generator=g)).to(torch.bfloat16).to(dev)
    xr = (x.float() * 3).to(torch.bfloat16)
    hn = torch.empty_like(xr)
    check(lib.merv_decode_rmsnorm(ptr(xr), ptr(wn), ptr(hn), 1, K, 1e-5, _st(dev)), "rms")
    two, fused = torch.empty_like(y), torch.empty_like(y)
    for W2p in (0, ptr(W2)):
        check(lib.merv_decode_gemv(ptr(W), W2p, ptr(hn), 0, ptr(two), 0, N, K, 0, 0.0, _st(dev)), "gemv")
        check(lib.merv_decode_gemv(ptr(W), W2p, ptr(xr), 0, ptr(fused), 0, N, K, ptr(wn), 1e-5, _st(dev)), "norm+gemv")
        assert torch.equal(fused, two)


@pytest.mark.parametrize("norm,gated", [(False, False), (True, False), (True, True), (False, True)])
def test_gemv_exact_and_general_instantiations_give_the_same_bits(dev, norm, gated):
    """decode.hip picks an EXACT instantiation (no per-chunk clamps / selects) when K / 8 is a multiple of the trip and N of the block's
    rows, the general one otherwise: the rows of a 4096 x 4096 launch (EXACT) must equal, bit for bit, the same rows computed as the head
    of a 4098-row launch (general: 4098 is no multiple of the block's rows)."""
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    N, K = 4096, 4096
    W = (torch.randn(N + 2, K, generator=g) * K**-0.5).to(torch.bfloat16).to(dev)
    W2 = (torch.randn(N + 2, K, generator=g) * K**-0.5).to(torch.bfloat16).to(dev)
    x = (torch.randn(K, generator=g) * 2).to(torch.bfloat16).to(dev)
    wn = (1 + 0.2 * torch.randn(K, generator=g)).to(torch.bfloat16).to(dev)
    ya = torch.empty(N, dtype=torch.bfloat16, device=dev)
    yb = torch.empty(N + 2, dtype=torch.bfloat16, device=dev)
    w2a, w2b = (ptr(W2[:N]), ptr(W2)) if gated else (0, 0)
    nw = ptr(wn) if norm else 0
    check(lib.merv_decode_gemv(ptr(W[:N]), w2a, ptr(x), 0, ptr(ya), 0, N, K, nw, 1e-5, _st(dev)), "exact")
    check(lib.merv_decode_gemv(ptr(W), w2b, ptr(x), 0, ptr(yb), 0, N + 2, K, nw, 1e-5, _st(dev)), "general")
    assert torch.equal(ya, yb[:N])


def test_gemv3_equals_three_gemvs(dev):
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(2)
    K = 4096
    Ns = (4096, 1024, 1026)
    Ws = [(torch.randn(n, K, generator=g) * K**-0.5).to(torch.bfloat16).to(dev) for n in Ns]
    x = torch.randn(K, generator=g).to(torch.bfloat16).to(dev)
    ys = [torch.empty(n, dtype=torch.bfloat16, device=dev) for n in Ns]
    check(lib.merv_decode_gemv3(ptr(Ws[0]), ptr(Ws[1]), ptr(Ws[2]), ptr(x), ptr(ys[0]), ptr(ys[1]), ptr(ys[2]), *Ns, K, 0, 0.0, _st(dev)), "gemv3")
    for W, y, n in zip(Ws, ys, Ns):
        one = torch.empty(n, dtype=torch.bfloat16, device=dev)
        check(lib.merv_decode_gemv(ptr(W), 0, ptr(x), 0, ptr(one), 0, n, K, 0, 0.0, _st(dev)), "gemv")
        assert torch.equal(y, one)


def _rot(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


@pytest.mark.parametrize("H,Hkv,pos", [(32, 32, 0), (32, 8, 5), (4, 1, 1048), (8, 8, 300), (2, 2, 7)])
def test_rope_cache_and_attention(dev, H, Hkv, pos):
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    hd, max_len, ns = 128, 1280, 8
    g = torch.Generator().manual_seed(H * 1000 + pos)
    bf = lambda t: t.to(torch.bfloat16).to(dev)
    q, k, v = bf(torch.randn(H * hd, generator=g)), bf(torch.randn(Hkv * hd, generator=g)), bf(torch.randn(Hkv * hd, generator=g))
    Kc, Vc = bf(torch.randn(Hkv, max_len, hd, generator=g)), bf(torch.randn(Hkv, max_len, hd, generator=g))
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.outer(torch.arange(max_len, dtype=torch.float32), inv)
    emb = torch.cat([fr, fr], -1)
    cos, sin = bf(emb.cos()), bf(emb.sin())
    p = torch.tensor([pos], dtype=torch.int64, device=dev)
    q2 = torch.empty_like(q)
    Kc0, Vc0 = Kc.clone(), Vc.clone()
    check(lib.merv_decode_rope_cache(ptr(q), ptr(k), ptr(v), ptr(q2), ptr(Kc), ptr(Vc), ptr(cos), ptr(sin), ptr(p), H, Hkv, hd, max_len,
                                     _st(dev)), "rope")
    qh, kh = q.view(H, hd), k.view(Hkv, hd)
    q_ref = qh * cos[pos] + _rot(qh) * sin[pos]  # bf16 tensor arithmetic, as apply_rotary_pos_emb on the bf16 module
    k_ref = kh * cos[pos] + _rot(kh) * sin[pos]
    assert torch.equal(q2.view(H, hd), q_ref) and torch.equal(Kc[:, pos], k_ref) and torch.equal(Vc[:, pos], v.view(Hkv, hd))
    keep = torch.ones(max_len, dtype=torch.bool, device=dev); keep[pos] = False
    assert torch.equal(Kc[:, keep], Kc0[:, keep]) and torch.equal(Vc[:, keep], Vc0[:, keep])  # nothing else touched
    out = torch.empty(H * hd, dtype=torch.bfloat16, device=dev)
    ws = torch.empty(lib.merv_decode_attention_workspace_floats(H, ns), dtype=torch.float32, device=dev)
    check(lib.merv_decode_attention(ptr(q2), ptr(Kc), ptr(Vc), ptr(out), ptr(ws), ptr(p), H, Hkv, hd, max_len, ns, hd**-0.5, _st(dev)), "attn")
    rep = H // Hkv
    Kf = Kc[:, : pos + 1].float().repeat_interleave(rep, 0)
    Vf = Vc[:, : pos + 1].float().repeat_interleave(rep, 0)
    att = torch.softmax((q2.view(H, 1, hd).float() @ Kf.transpose(1, 2)) * hd**-0.5, -1)
    ref = (att @ Vf).reshape(-1)
    assert torch.isfinite(out.float()).all()
    assert rel_l2(out, ref) < 6e-3


@pytest.mark.parametrize("H,Hkv,pos", [(32, 32, 0), (32, 8, 5), (4, 1, 1048), (8, 8, 300), (2, 2, 7), (32, 32, 1111), (32, 8, 15), (32, 8, 16)])
def test_fused_attention_launch_equals_the_three_kernels(dev, H, Hkv, pos):
    """merv_decode_attention_fused (rotary + cache write + split attention + last-arrival merge in one launch) gives the bits
    of merv_decode_rope_cache -> merv_decode_attention, leaves the same cache behind, and its arrival counters return to zero
    (three back-to-back launches on the same workspace, as a replayed graph would issue them)."""
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    hd, max_len, ns = 128, 1280, 8
    g = torch.Generator().manual_seed(H * 1000 + pos + 1)
    bf = lambda t: t.to(torch.bfloat16).to(dev)
    q, k, v = bf(torch.randn(H * hd, generator=g)), bf(torch.randn(Hkv * hd, generator=g)), bf(torch.randn(Hkv * hd, generator=g))
    Kc, Vc = bf(torch.randn(Hkv, max_len, hd, generator=g)), bf(torch.randn(Hkv, max_len, hd, generator=g))
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    emb = torch.outer(torch.arange(max_len, dtype=torch.float32), inv)
    emb = torch.cat([emb, emb], -1)
    cos, sin = bf(emb.cos()), bf(emb.sin())
    p = torch.tensor([pos], dtype=torch.int64, device=dev)
    Ka, Va, q2 = Kc.clone(), Vc.clone(), torch.empty_like(q)
    out_a = torch.empty(H * hd, dtype=torch.bfloat16, device=dev)
    ws_a = torch.empty(lib.merv_decode_attention_workspace_floats(H, ns), dtype=torch.float32, device=dev)
    check(lib.merv_decode_rope_cache(ptr(q), ptr(k), ptr(v), ptr(q2), ptr(Ka), ptr(Va), ptr(cos), ptr(sin), ptr(p), H, Hkv, hd, max_len,
                                     _st(dev)), "rope")
    check(lib.merv_decode_attention(ptr(q2), ptr(Ka), ptr(Va), ptr(out_a), ptr(ws_a), ptr(p), H, Hkv, hd, max_len, ns, hd**-0.5, _st(dev)), "attn")
    n_ws = lib.merv_decode_attention_fused_workspace_floats(H, ns)
    assert n_ws == H * ns * 130 + 32 * H
    ws = torch.zeros(n_ws, dtype=torch.float32, device=dev)
    for rep in range(3):
        Kb, Vb = Kc.clone(), Vc.clone()
        out_b = torch.full((H * hd,), float("nan"), dtype=torch.bfloat16, device=dev)
        check(lib.merv_decode_attention_fused(ptr(q), ptr(k), ptr(v), ptr(cos), ptr(sin), ptr(p), ptr(Kb), ptr(Vb), ptr(out_b), ptr(ws),
                                              H, Hkv, hd, max_len, ns, hd**-0.5, _st(dev)), "fused")
        torch.cuda.synchronize()
        assert torch.equal(out_b, out_a), rep
        assert torch.equal(Kb, Ka) and torch.equal(Vb, Va)
        assert int(ws[H * ns * 130:].view(torch.int32).abs().sum()) == 0  # counters back at zero
    # one split (the ticket drawn is at once the last): another summation order, so against the fp32 restatement
    Kb, Vb = Kc.clone(), Vc.clone()
    out_c = torch.full((H * hd,), float("nan"), dtype=torch.bfloat16, device=dev)
    ws = torch.zeros(lib.merv_decode_attention_fused_workspace_floats(H, 1), dtype=torch.float32, device=dev)  # the counters sit behind H * nsplit partials
    check(lib.merv_decode_attention_fused(ptr(q), ptr(k), ptr(v), ptr(cos), ptr(sin), ptr(p), ptr(Kb), ptr(Vb), ptr(out_c), ptr(ws),
                                          H, Hkv, hd, max_len, 1, hd**-0.5, _st(dev)), "fused1")
    torch.cuda.synchronize()
    assert torch.equal(Kb, Ka) and torch.equal(Vb, Va)
    rep = H // Hkv
    Kf = Ka[:, : pos + 1].float().repeat_interleave(rep, 0)
    Vf = Va[:, : pos + 1].float().repeat_interleave(rep, 0)
    att = torch.softmax((q2.view(H, 1, hd).float() @ Kf.transpose(1, 2)) * hd**-0.5, -1)
    ref = (att @ Vf).reshape(-1)
    assert rel_l2(out_c, ref) < 6e-3 and rel_l2(out_c, out_a) < 6e-3


@pytest.mark.parametrize("kv_heads,family", [(2, "llama"), (1, "mistral"), (1, "llama3.1"), (1, "qwen2")])
def test_hip_decoder_matches_pytorch_decoder(dev, kv_heads, family):
    """Same random model, same prefill (PyTorch-ROCm, as the north_star keeps it), then token-by-token decode on both decoders,
    both fed the PyTorch decoder's greedy tokens: logits within the bf16 tolerance at every step, graph replay == eager."""
    from merv_amd.llm import HipDecoder, LlamaBackbone, StaticDecoder
    cfg = dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
               num_key_value_heads=kv_heads, max_position_embeddings=2048, bos_token_id=1, eos_token_id=2, pad_token_id=0)
    if family == "mistral":
        cfg.update(rope_theta=1e6, sliding_window=None)
    if family == "llama3.1":  # GQA + the "llama3" frequency-dependent rotary scaling (tables from the module's rotary embedding)
        cfg.update(rope_theta=500000.0, rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0,
                                                      "high_freq_factor": 4.0, "original_max_position_embeddings": 64})
        family = "llama"
    if family == "qwen2":  # q / k / v biases (merv_decode_gemv3_bias), tied embeddings
        cfg.update(rope_theta=1e6, tie_word_embeddings=True, use_sliding_window=False)
    llm = LlamaBackbone(cfg, device=dev, family=family)
    if family == "qwen2":
        with torch.no_grad():
            for lyr in llm.llm.model.layers:  # HF initialises the biases to zero: make them matter
                for pr in (lyr.self_attn.q_proj, lyr.self_attn.k_proj, lyr.self_attn.v_proj):
                    pr.bias.normal_(0, 0.5)
    assert HipDecoder.supports(llm.llm, 1) and not HipDecoder.supports(llm.llm, 2)
    ref, hip, hip_eager = StaticDecoder(llm.llm, 256, 1), HipDecoder(llm.llm, 256, 1), HipDecoder(llm.llm, 256, 1)
    hip.use_hip_prefill = hip_eager.use_hip_prefill = False  # the decode steps alone: all three start from the same PyTorch prefill
    emb = (torch.randn(1, 37, 256, generator=torch.Generator().manual_seed(3)) * 0.5).to(torch.bfloat16).to(dev)
    l_ref, l_hip = ref.prefill(emb), hip.prefill(emb)
    hip_eager.prefill(emb)
    assert torch.equal(l_ref, l_hip)  # the prefill is the same PyTorch code
    worst = 0.0
    for step in range(12):
        tok = l_ref.argmax(-1)
        l_ref = ref.decode(tok, use_graph=False).clone()
        l_hip = hip.decode(tok, use_graph=True).clone()
        l_e = hip_eager.decode(tok, use_graph=False).clone()
        assert torch.equal(l_hip, l_e), step  # replayed graph == the same kernels launched eagerly
        err = rel_l2(l_hip, l_ref)
        worst = max(worst, err)
        assert err < 2e-2, (step, err)
        top2 = l_ref[0].topk(2).values
        if float(top2[0] - top2[1]) > 0.05 * float(l_ref[0].abs().max()):  # a clear winner: both decoders pick it
            assert int(l_hip.argmax(-1)) == int(l_ref.argmax(-1))
    print(f"HipDecoder vs StaticDecoder ({family}, kv heads {kv_heads}): worst logits rel-L2 {worst:.3e}")


def test_prefill_rope_cache_rows(dev):
    """merv_prefill_rope_cache == the module's expression q * cos + rotate_half(q) * sin on bf16 tensors, bit for bit, q in place,
    k / v into the cache rows pos0 .. pos0 + S - 1 (other rows untouched)."""
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    S, H, Hkv, hd, max_len, pos0 = 37, 4, 2, 128, 64, 5
    bf = lambda t: t.to(torch.bfloat16).to(dev)
    q, k, v = bf(torch.randn(S, H * hd, generator=g)), bf(torch.randn(S, Hkv * hd, generator=g)), bf(torch.randn(S, Hkv * hd, generator=g))
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    emb = torch.outer(torch.arange(max_len, dtype=torch.float32), inv)
    emb = torch.cat([emb, emb], -1)
    cos, sin = bf(emb.cos()), bf(emb.sin())
    rot = lambda x: torch.cat([-x[..., hd // 2:], x[..., :hd // 2]], -1)
    c, s_ = cos[pos0:pos0 + S][:, None], sin[pos0:pos0 + S][:, None]
    q4, k4 = q.view(S, H, hd), k.view(S, Hkv, hd)
    q_ref = q4 * c + rot(q4) * s_
    k_ref = k4 * c + rot(k4) * s_
    Kc = bf(torch.randn(Hkv, max_len, hd, generator=g))
    Vc = bf(torch.randn(Hkv, max_len, hd, generator=g))
    K0, V0 = Kc.clone(), Vc.clone()
    qkv = torch.cat([q, k, v], 1).contiguous()  # the same through column ranges of one [S, (H + 2 Hkv) * hd] buffer
    check(lib.merv_prefill_rope_cache(ptr(q), ptr(k), ptr(v), ptr(Kc), ptr(Vc), ptr(cos), ptr(sin), S, pos0, H, Hkv, hd, max_len,
                                      H * hd, Hkv * hd, _st(dev)), "rope")
    assert torch.equal(q.view(S, H, hd), q_ref)
    K2, V2 = K0.clone(), V0.clone()
    ld = (H + 2 * Hkv) * hd
    check(lib.merv_prefill_rope_cache(ptr(qkv), qkv.data_ptr() + 2 * H * hd, qkv.data_ptr() + 2 * (H + Hkv) * hd, ptr(K2), ptr(V2), ptr(cos),
                                      ptr(sin), S, pos0, H, Hkv, hd, max_len, ld, ld, _st(dev)), "rope strided")
    assert torch.equal(qkv[:, :H * hd], q) and torch.equal(K2, Kc) and torch.equal(V2, Vc)
    assert torch.equal(Kc[:, pos0:pos0 + S], k_ref.transpose(0, 1)) and torch.equal(Vc[:, pos0:pos0 + S], v.view(S, Hkv, hd).transpose(0, 1))
    keep = torch.ones(max_len, dtype=torch.bool)
    keep[pos0:pos0 + S] = False
    assert torch.equal(Kc[:, keep], K0[:, keep]) and torch.equal(Vc[:, keep], V0[:, keep])


@pytest.mark.parametrize("H,Hkv,pos,D", [(32, 32, 1049, 4096), (32, 8, 300, 4096), (8, 2, 5, 1000), (40, 40, 77, 5120), (4, 4, 0, 512)])
def test_split_attention_and_merging_oproj_equal_the_fused_pair(dev, H, Hkv, pos, D):
    """merv_decode_attention_split + merv_decode_oproj_merge == merv_decode_attention_fused + merv_decode_gemv(o_proj, residual), bit
    for bit: same caches, same merged attention vector, same x -- and x may be updated in place."""
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    hd, max_len, ns = 128, 1280, 8
    g = torch.Generator().manual_seed(H + pos)
    bf = lambda t: t.to(torch.bfloat16).to(dev)
    Kc, Vc = bf(torch.randn(Hkv, max_len, hd, generator=g)), bf(torch.randn(Hkv, max_len, hd, generator=g))
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    emb = torch.outer(torch.arange(max_len, dtype=torch.float32), inv)
    emb = torch.cat([emb, emb], -1)
    cos, sin = bf(emb.cos()), bf(emb.sin())
    q, k, v = bf(torch.randn(H * hd, generator=g)), bf(torch.randn(Hkv * hd, generator=g)), bf(torch.randn(Hkv * hd, generator=g))
    Wo = bf(torch.randn(D, H * hd, generator=g) * (H * hd) ** -0.5)
    x0 = bf(torch.randn(D, generator=g))
    p = torch.tensor([pos], dtype=torch.int64, device=dev)
    st = _st(dev)
    ws = torch.zeros(lib.merv_decode_attention_fused_workspace_floats(H, ns), dtype=torch.float32, device=dev)
    Ka, Va, out_a, xa = Kc.clone(), Vc.clone(), torch.empty(H * hd, dtype=torch.bfloat16, device=dev), x0.clone()
    check(lib.merv_decode_attention_fused(ptr(q), ptr(k), ptr(v), ptr(cos), ptr(sin), ptr(p), ptr(Ka), ptr(Va), ptr(out_a), ptr(ws), H, Hkv, hd,
                                          max_len, ns, hd**-0.5, st), "fused")
    check(lib.merv_decode_gemv(ptr(Wo), 0, ptr(out_a), ptr(xa), ptr(xa), 0, D, H * hd, 0, 0.0, st), "o_proj")
    ws2 = torch.full((lib.merv_decode_attention_split_workspace_floats(H, ns),), float("nan"), dtype=torch.float32, device=dev)
    Kb, Vb, out_b, xb = Kc.clone(), Vc.clone(), torch.empty(H * hd, dtype=torch.bfloat16, device=dev), x0.clone()
    check(lib.merv_decode_attention_split(ptr(q), ptr(k), ptr(v), ptr(cos), ptr(sin), ptr(p), ptr(Kb), ptr(Vb), ptr(ws2), H, Hkv, hd, max_len,
                                          ns, hd**-0.5, st), "split")
    check(lib.merv_decode_oproj_merge(ptr(Wo), ptr(xb), ptr(xb), ptr(ws2), ptr(out_b), D, H, hd, ns, st), "oproj_merge")
    assert torch.equal(Ka, Kb) and torch.equal(Va, Vb)
    assert torch.equal(out_a, out_b)
    assert torch.equal(xa, xb)


def test_fused_attention_handoff_stress(dev):
    """The in-launch hand-off of merv_decode_attention_fused (split partials -> write-through stores -> vmcnt(0) -> ticket -> the last
    block merges) under timing noise: 400 launches at random positions, two back to back on the same workspace each time, with a second
    stream hammering HBM every third iteration -- every one bit-equal to the three-kernel sequence, counters back at zero. (The
    3000-iteration form is tools/probes/decode_attention_stress.py.)"""
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    H, Hkv, hd, max_len, ns = 32, 8, 128, 2048, 8
    g = torch.Generator().manual_seed(1)
    bf = lambda t: t.to(torch.bfloat16).to(dev)
    Kc, Vc = bf(torch.randn(Hkv, max_len, hd, generator=g)), bf(torch.randn(Hkv, max_len, hd, generator=g))
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    emb = torch.outer(torch.arange(max_len, dtype=torch.float32), inv)
    emb = torch.cat([emb, emb], -1)
    cos, sin = bf(emb.cos()), bf(emb.sin())
    ws = torch.zeros(lib.merv_decode_attention_fused_workspace_floats(H, ns), dtype=torch.float32, device=dev)
    ws_a = torch.empty(lib.merv_decode_attention_workspace_floats(H, ns), dtype=torch.float32, device=dev)
    side = torch.cuda.Stream(dev)
    big = torch.randn(32, 1024, 1024, device=dev)
    st = _st(dev)
    bad = 0
    for it in range(400):
        pos = int(torch.randint(0, max_len - 1, (1,), generator=g))
        q, k, v = bf(torch.randn(H * hd, generator=g)), bf(torch.randn(Hkv * hd, generator=g)), bf(torch.randn(Hkv * hd, generator=g))
        p = torch.tensor([pos], dtype=torch.int64, device=dev)
        Ka, Va, q2 = Kc.clone(), Vc.clone(), torch.empty_like(q)
        out_a = torch.empty(H * hd, dtype=torch.bfloat16, device=dev)
        check(lib.merv_decode_rope_cache(ptr(q), ptr(k), ptr(v), ptr(q2), ptr(Ka), ptr(Va), ptr(cos), ptr(sin), ptr(p), H, Hkv, hd, max_len, st), "rope")
        check(lib.merv_decode_attention(ptr(q2), ptr(Ka), ptr(Va), ptr(out_a), ptr(ws_a), ptr(p), H, Hkv, hd, max_len, ns, hd**-0.5, st), "attn")
        Kb, Vb = Kc.clone(), Vc.clone()
        out_b = torch.full((H * hd,), float("nan"), dtype=torch.bfloat16, device=dev)
        if it % 3 == 0:
            with torch.cuda.stream(side):
                big.mul_(1.0001)
        for _ in range(2):
            check(lib.merv_decode_attention_fused(ptr(q), ptr(k), ptr(v), ptr(cos), ptr(sin), ptr(p), ptr(Kb), ptr(Vb), ptr(out_b), ptr(ws),
                                                  H, Hkv, hd, max_len, ns, hd**-0.5, st), "fused")
        bad += int(not (torch.equal(out_a, out_b) and torch.equal(Ka, Kb) and torch.equal(Va, Vb)))
    torch.cuda.synchronize(dev)
    assert bad == 0
    assert int(ws[H * ns * 130:].view(torch.int32).abs().sum()) == 0  # every head's arrival counter restored


@pytest.mark.parametrize("S,H,Hkv", [(1, 2, 2), (37, 4, 2), (128, 2, 1), (129, 2, 2), (449, 4, 4), (1049, 8, 2)])
def test_prefill_attention_vs_sdpa(dev, S, H, Hkv):
    """merv_prefill_attention (causal, head dim 128, K / V read from the cache layout, output in the o-projection's layout) against
    F.scaled_dot_product_attention(is_causal=True) in fp32 on the same bf16 inputs."""
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(100 + S)
    hd, max_len = 128, 1280
    bf = lambda t: t.to(torch.bfloat16).to(dev)
    q = bf(torch.randn(S, H * hd, generator=g) * 1.5)
    Kc, Vc = bf(torch.randn(Hkv, max_len, hd, generator=g)), bf(torch.randn(Hkv, max_len, hd, generator=g))
    if S > 40:  # a query with one dominant late key and one with a dominant early key: the running maximum has to move / hold
        q[S - 3, :hd] = (Kc[0, S - 5] * 0.6).to(q.dtype)
        q[S // 2, :hd] = (Kc[0, 1] * 0.6).to(q.dtype)
    out = torch.full((S, H * hd), float("nan"), dtype=torch.bfloat16, device=dev)
    check(lib.merv_prefill_attention(ptr(q), ptr(Kc), ptr(Vc), ptr(out), S, H, Hkv, hd, H * hd, hd, max_len * hd, H * hd, hd**-0.5, _st(dev)),
          "merv_prefill_attention")
    qf = q.float().view(1, S, H, hd).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qf, Kc[None, :, :S].float(), Vc[None, :, :S].float(), is_causal=True, enable_gqa=H != Hkv)
    ref = ref.transpose(1, 2).reshape(S, H * hd)
    assert torch.isfinite(out.float()).all()
    err = rel_l2(out, ref)
    worst_row = float(((out.float() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-6)).max())
    print(f"prefill attention S={S} H={H}/{Hkv}: rel-L2 {err:.3e}, worst row {worst_row:.3e}")
    assert err < 8e-3 and worst_row < 3e-2


@pytest.mark.parametrize("rows,D", [(5, 256), (77, 4096), (9, 5120), (3, 8192)])
def test_add_rmsnorm_equals_add_then_rmsnorm(dev, rows, D):
    """merv_add_rmsnorm == (x + delta in bf16) followed by merv_decode_rmsnorm, bit for bit; x is updated in place."""
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(rows + D)
    bf = lambda t: t.to(torch.bfloat16).to(dev)
    x, d, w = bf(torch.randn(rows, D, generator=g)), bf(torch.randn(rows, D, generator=g) * 0.3), bf(torch.randn(D, generator=g))
    xs = x + d
    y_ref = torch.empty_like(xs)
    check(lib.merv_decode_rmsnorm(ptr(xs), ptr(w), ptr(y_ref), rows, D, 1e-5, _st(dev)), "rmsnorm")
    y = torch.empty_like(x)
    check(lib.merv_add_rmsnorm(ptr(x), ptr(d), ptr(w), ptr(y), rows, D, 1e-5, _st(dev)), "add_rmsnorm")
    assert torch.equal(x, xs) and torch.equal(y, y_ref)


def test_silu_mul(dev):
    """merv_silu_mul == F.silu(gate) * up on bf16 tensors (in place into gate as well)."""
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(12)
    gate = (torch.randn(333, 1376, generator=g) * 3).to(torch.bfloat16).to(dev)
    up = torch.randn(333, 1376, generator=g).to(torch.bfloat16).to(dev)
    ref = F.silu(gate) * up
    out = torch.empty_like(gate)
    check(lib.merv_silu_mul(ptr(gate), ptr(up), ptr(out), gate.numel(), _st(dev)), "silu_mul")
    diff = (out.float() - ref.float()).abs()
    ulp = ref.float().abs() * 2.0 ** -7 + 1e-30
    assert float((diff / ulp).max()) <= 1.0 and float((diff > 0).float().mean()) < 1e-3  # exp() implementations differ in the last place, rarely
    check(lib.merv_silu_mul(ptr(gate), ptr(up), ptr(gate), gate.numel(), _st(dev)), "silu_mul in place")
    assert torch.equal(gate, out)


@pytest.mark.parametrize("kv_heads,family", [(2, "llama"), (1, "llama3.1"), (1, "qwen2")])
def test_hip_prefill_matches_pytorch_prefill(dev, kv_heads, family):
    """HipDecoder's prefill (RMSNorm / rotary + cache fill / silu * up on the library's kernels, GEMMs and attention on PyTorch-ROCm)
    against StaticDecoder's plain PyTorch expression: same rounding points, so caches and logits agree to bf16 rounding noise."""
    from merv_amd.llm import HipDecoder, LlamaBackbone, StaticDecoder
    cfg = dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
               num_key_value_heads=kv_heads, max_position_embeddings=2048, bos_token_id=1, eos_token_id=2, pad_token_id=0)
    if family == "llama3.1":
        cfg.update(rope_theta=500000.0, rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0,
                                                      "high_freq_factor": 4.0, "original_max_position_embeddings": 64})
        family = "llama"
    if family == "qwen2":
        cfg.update(rope_theta=1e6, tie_word_embeddings=True, use_sliding_window=False)
    llm = LlamaBackbone(cfg, device=dev, family=family)
    if family == "qwen2":
        with torch.no_grad():
            for lyr in llm.llm.model.layers:
                for pr in (lyr.self_attn.q_proj, lyr.self_attn.k_proj, lyr.self_attn.v_proj):
                    pr.bias.normal_(0, 0.5)
    ref, hip = StaticDecoder(llm.llm, 256, 1), HipDecoder(llm.llm, 256, 1)
    hip.use_hip_prefill = hip.use_hip_prefill_attn = True  # (the defaults; MERV_HIP_PREFILL=0 in the environment would switch them off)
    emb = (torch.randn(1, 77, 256, generator=torch.Generator().manual_seed(4)) * 0.5).to(torch.bfloat16).to(dev)
    emb0 = emb.clone()
    sd0 = {k: v.clone() for k, v in llm.llm.state_dict().items()}
    l_ref, l_hip = ref.prefill(emb), hip.prefill(emb)
    # the q / k / v parameters now live in one fused matrix per layer (one GEMM): same values, contiguous, still ordinary tensors
    for k, v in llm.llm.state_dict().items():
        assert torch.equal(v, sd0[k]), k
    a0 = llm.llm.model.layers[0].self_attn
    assert a0.k_proj.weight.data_ptr() == a0.q_proj.weight.data_ptr() + a0.q_proj.weight.numel() * 2
    assert a0.v_proj.weight.data_ptr() == a0.k_proj.weight.data_ptr() + a0.k_proj.weight.numel() * 2
    assert not a0.q_proj.weight.is_inference() and a0.q_proj.weight.is_contiguous()
    with torch.no_grad():  # an in-place load reaches the fused matrix; a replaced storage is noticed and fused again
        a0.q_proj.weight.mul_(1.0)
    fused_before = a0._merv_qkv[0]
    hip.prefill(emb)
    assert a0._merv_qkv[0] is fused_before
    a0.k_proj.weight.data = a0.k_proj.weight.data.clone()
    l_again = hip.prefill(emb)
    assert a0._merv_qkv[0] is not fused_before and torch.equal(l_again, l_hip)
    assert torch.equal(emb, emb0)  # the caller's embeddings are not the in-place residual stream
    assert int(hip.pos) == 77 == int(ref.pos)
    assert rel_l2(l_hip, l_ref) < 2e-2
    for li in range(3):
        assert rel_l2(hip.K[li][:, :, :77], ref.K[li][:, :, :77]) < 2e-2 and rel_l2(hip.V[li][:, :, :77], ref.V[li][:, :, :77]) < 2e-2
        assert float(hip.K[li][:, :, 77:].abs().max()) == 0.0
    assert torch.equal(hip.K[0][:, :, :77], ref.K[0][:, :, :77])  # layer 0: same inputs, same rounding points (RMSNorm sums aside)
    # decoding continues from either prefill
    tok = l_ref.argmax(-1)
    a, b = ref.decode(tok, use_graph=False), hip.decode(tok, use_graph=False)
    assert rel_l2(b, a) < 2e-2


def test_greedy_advance_kernel_follows_torch_argmax(dev):
    """merv_decode_greedy_advance: token = torch.argmax(logits) -- first of tied maxima, a NaN wins --, logged at out[pos - pos0], pos + 1."""
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    for case in range(6):
        V = [32000, 32064, 320, 1, 151936, 5000][case]
        lg = torch.randn(V, generator=g)
        if case == 1:
            lg[[17, 9000, 31000]] = 9.0       # tied maxima: the first wins
        if case == 2:
            lg[:] = -float("inf")              # all equal
        if case == 5:
            lg[[4000, 77]] = float("nan")      # NaN wins, the first one
        lg = lg.to(dev)
        tok = torch.full((1, 1), -1, dtype=torch.long, device=dev)
        pos = torch.tensor([40 + case], dtype=torch.long, device=dev)
        out = torch.full((64,), -1, dtype=torch.long, device=dev)
        check(lib.merv_decode_greedy_advance(ptr(lg), V, ptr(tok), ptr(pos), ptr(out), 38, _st(dev)), "greedy_advance")
        want = int(torch.argmax(lg))
        assert int(tok) == want and int(pos) == 41 + case and int(out[2 + case]) == want and int((out != -1).sum()) == 1, case


@pytest.mark.parametrize("eos", [None, "hit"])
def test_greedy_generation_on_the_device_equals_the_host_loop(dev, eos):
    """generate_from_embeds(do_sample=False) with the argmax / token hand-over / position increment inside the captured step
    (HipDecoder.greedy_run) returns the host loop's tokens, with and without an EOS that stops it."""
    from merv_amd.llm import HipDecoder, LlamaBackbone
    llm = LlamaBackbone(dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                             num_key_value_heads=2, max_position_embeddings=2048, bos_token_id=1, eos_token_id=None, pad_token_id=0), device=dev)
    emb = (torch.randn(1, 20, 256, generator=torch.Generator().manual_seed(7)) * 0.5).to(torch.bfloat16).to(dev)
    was = HipDecoder.use_greedy_graph
    try:
        HipDecoder.use_greedy_graph = False
        ref = llm.generate_from_embeds(emb, max_new_tokens=21)
        llm._decoders.clear()
        HipDecoder.use_greedy_graph = True
        eos_id = int(ref[0, 10]) if eos else None
        HipDecoder.use_greedy_graph = False
        ref = llm.generate_from_embeds(emb, max_new_tokens=21, eos_token_id=eos_id)
        llm._decoders.clear()
        HipDecoder.use_greedy_graph = True
        out = llm.generate_from_embeds(emb, max_new_tokens=21, eos_token_id=eos_id)
        out2 = llm.generate_from_embeds(emb, max_new_tokens=21, eos_token_id=eos_id)  # the captured graph again, from a fresh prefill
    finally:
        HipDecoder.use_greedy_graph = was
    assert isinstance(next(iter(llm._decoders.values())), HipDecoder)
    assert torch.equal(out, ref) and torch.equal(out2, ref)
    if eos:
        assert int(out[0, -1]) == eos_id and out.shape[1] <= 11


def test_one_decoder_alternating_greedy_and_sampled_generations(dev):
    """ONE HipDecoder serving greedy (device-side greedy_run graphs) and non-greedy (decode() graph) generations in turn, both orders:
    the two sets of captured graphs embed from the same token tensor, which is never rebound (a rebound tensor left the
    other path's graph reading freed memory). Greedy results must equal those of a fresh decoder; sampled ones must equal the same
    seeded call on a fresh decoder; a first token that already is the EOS returns at once."""
    from merv_amd.llm import HipDecoder, LlamaBackbone
    llm = LlamaBackbone(dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                             num_key_value_heads=2, max_position_embeddings=2048, bos_token_id=1, eos_token_id=None, pad_token_id=0), device=dev)
    emb = (torch.randn(1, 20, 256, generator=torch.Generator().manual_seed(11)) * 0.5).to(torch.bfloat16).to(dev)
    gen = lambda seed: torch.Generator(device=dev).manual_seed(seed)
    sample = lambda: llm.generate_from_embeds(emb, max_new_tokens=19, do_sample=True, temperature=0.9, top_k=40, generator=gen(5))
    penal = lambda: llm.generate_from_embeds(emb, max_new_tokens=19, repetition_penalty=1.3)
    greedy = lambda: llm.generate_from_embeds(emb, max_new_tokens=19)
    ref = {}
    for name, fn in (("greedy", greedy), ("sample", sample), ("penal", penal)):  # each on a fresh decoder
        getattr(llm, "_decoders", {}).clear()
        ref[name] = fn()
    for order in (("greedy", "sample", "greedy", "penal", "greedy", "sample"), ("sample", "greedy", "penal", "greedy")):
        llm._decoders.clear()
        fns = {"greedy": greedy, "sample": sample, "penal": penal}
        dec0 = None
        for i, name in enumerate(order):
            out = fns[name]()
            dec = next(iter(llm._decoders.values()))
            assert isinstance(dec, HipDecoder) and (dec0 is None or dec is dec0)  # one decoder throughout
            dec0 = dec
            assert torch.equal(out, ref[name]), (order, i, name, out.tolist(), ref[name].tolist())
        assert dec0.tok.data_ptr() == dec.tok.data_ptr() and dec0.graph is not None and dec0.greedy_graph is not None
    first = int(ref["greedy"][0, 0])
    only = llm.generate_from_embeds(emb, max_new_tokens=19, eos_token_id=first)
    assert only.tolist() == [[first]]
    assert llm.generate_from_embeds(emb, max_new_tokens=1).tolist() == [[first]]


def _sample_params(dev, temperature, seed, eos=-1, min_new=0, first_pos=0):
    import struct
    blob = struct.pack("<fiQqq", 1.0 / temperature, eos, seed, min_new, first_pos)
    return torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)


@pytest.mark.parametrize("V,temperature", [(13, 0.4), (320, 1.0), (32064, 0.7)])
def test_sample_advance_kernel_draws_from_softmax_of_logits_over_temperature(dev, V, temperature):
    """merv_decode_sample_advance (round 6: `do_sample=True, temperature=T` of scripts/quick_start.py:24-32 inside the captured step): the token
    is argmax(logits / T + Gumbel noise from a Philox stream keyed by (seed, position)), i.e. a draw from softmax(logits / T). Over 40 000
    positions the token frequencies must match those probabilities -- and torch.multinomial's own frequencies on the same distribution --
    within sampling error; (seed, position) is reproducible, another seed is another stream, a NaN logit is never drawn."""
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(V)
    lg = torch.randn(V, generator=g) * 2.0
    if V > 1000:  # a realistic head: a few likely tokens, a long tail
        lg[:8] += 6.0
    lg[V // 2] = float("nan")
    probs = torch.softmax(torch.nan_to_num(lg, nan=-float("inf")).double() / temperature, -1)
    lgd = lg.to(dev)
    N = 40000
    tok = torch.zeros(1, 1, dtype=torch.long, device=dev)

    def run(seed, n=N, pos0=100):
        pos = torch.tensor([pos0], dtype=torch.long, device=dev)
        out = torch.full((n,), -1, dtype=torch.long, device=dev)
        prm = _sample_params(dev, temperature, seed)
        for _ in range(n):
            check(lib.merv_decode_sample_advance(ptr(lgd), V, ptr(prm), ptr(tok), ptr(pos), ptr(out), pos0, _st(dev)), "sample_advance")
        torch.cuda.synchronize()
        assert int(pos) == pos0 + n and int(tok) == int(out[-1])
        return out.cpu()

    out = run(1234)
    assert int(out.min()) >= 0 and int(out.max()) < V and not bool((out == V // 2).any())
    # frequencies over three seeds pooled (a single stream of 40 000 draws sits up to 4 sigma off in one bin now and then, like any sample:
    # seed 1234 does at V = 13 -- the same arithmetic on the CPU gives the same counts --, the pool does not)
    pooled = torch.cat([out, run(11), run(22)])
    N = pooled.numel()
    freq = torch.bincount(pooled, minlength=V).double() / N
    ref = torch.bincount(torch.multinomial(probs.float(), N, replacement=True, generator=torch.Generator().manual_seed(3)), minlength=V).double() / N
    sigma = torch.sqrt(probs * (1 - probs) / N)
    top = probs.topk(min(V, 12)).indices  # where the mass is: per-token frequencies within 5 sigma of the probability
    assert bool(((freq[top] - probs[top]).abs() <= 5 * sigma[top] + 1e-4).all()), (freq[top], probs[top])
    # the whole distribution: total variation against the exact probabilities no worse than torch.multinomial's own sample (+ slack)
    tv, tv_ref = float((freq - probs).abs().sum() / 2), float((ref - probs).abs().sum() / 2)
    assert tv <= 2.0 * tv_ref + 5e-3, (tv, tv_ref)
    assert torch.equal(run(1234, 512), out[:512])            # (seed, position) reproduces
    assert not torch.equal(run(99, 512), out[:512])          # another seed: another stream
    assert not torch.equal(run(1234, 512, pos0=101)[:-1], out[:511]) or V < 20  # (positions shifted by one: other numbers)


def test_sample_advance_bars_the_eos_token_until_min_new_tokens(dev):
    from merv_amd import _lib
    from merv_amd._lib import check, ptr
    lib = _lib.load()
    V = 64
    lg = torch.zeros(V)
    lg[5] = 30.0  # the end-of-sequence token, overwhelmingly likely
    lgd = lg.to(dev)
    tok = torch.zeros(1, 1, dtype=torch.long, device=dev)
    pos = torch.tensor([50], dtype=torch.long, device=dev)
    out = torch.full((16,), -1, dtype=torch.long, device=dev)
    prm = _sample_params(dev, 1.0, 7, eos=5, min_new=6, first_pos=50)  # the step at position 50 draws new token number 1
    for _ in range(16):
        check(lib.merv_decode_sample_advance(ptr(lgd), V, ptr(prm), ptr(tok), ptr(pos), ptr(out), 50, _st(dev)), "sample_advance")
    o = out.cpu().tolist()
    assert all(t != 5 for t in o[:5]) and all(t == 5 for t in o[5:]), o  # numbers 1..5 barred, number 6 onwards free


def test_sampled_generation_on_the_device(dev):
    """generate_from_embeds(do_sample=True, temperature=T) without top-k / top-p / penalty runs on HipDecoder.sample_run: one pair of graphs
    for every temperature / seed (they are read from device memory), a seeded generator reproduces the tokens, another seed does not, an EOS
    cuts the result, and greedy / host-loop generations on the same decoder are unaffected."""
    from merv_amd.llm import HipDecoder, LlamaBackbone
    llm = LlamaBackbone(dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                             num_key_value_heads=2, max_position_embeddings=2048, bos_token_id=1, eos_token_id=None, pad_token_id=0), device=dev)
    emb = (torch.randn(1, 20, 256, generator=torch.Generator().manual_seed(21)) * 0.5).to(torch.bfloat16).to(dev)
    gen = lambda seed: torch.Generator(device=dev).manual_seed(seed)
    greedy = llm.generate_from_embeds(emb, max_new_tokens=33)
    a = llm.generate_from_embeds(emb, max_new_tokens=33, do_sample=True, temperature=2.0, generator=gen(5))
    dec = next(iter(llm._decoders.values()))
    assert isinstance(dec, HipDecoder) and dec.sample_graph is not None and a.shape == (1, 33)
    b = llm.generate_from_embeds(emb, max_new_tokens=33, do_sample=True, temperature=2.0, generator=gen(5))
    c = llm.generate_from_embeds(emb, max_new_tokens=33, do_sample=True, temperature=2.0, generator=gen(6))
    cold = llm.generate_from_embeds(emb, max_new_tokens=33, do_sample=True, temperature=1e-3, generator=gen(6))  # T -> 0: the greedy text
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.equal(cold, greedy), (cold.tolist(), greedy.tolist())
    assert next(iter(llm._decoders.values())) is dec  # same decoder, same graphs
    eos_id = int(a[0, 12])
    cut = llm.generate_from_embeds(emb, max_new_tokens=33, do_sample=True, temperature=2.0, generator=gen(5), eos_token_id=eos_id)
    k = a[0].tolist().index(eos_id)
    assert cut.tolist() == [a[0, : k + 1].tolist()]
    assert torch.equal(llm.generate_from_embeds(emb, max_new_tokens=33), greedy)


def test_generate_uses_hip_decoder_when_it_can(dev):
    from merv_amd.llm import HipDecoder, LlamaBackbone, StaticDecoder
    llm = LlamaBackbone(dict(vocab_size=320, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                             num_key_value_heads=2, max_position_embeddings=2048, bos_token_id=1, eos_token_id=None, pad_token_id=0), device=dev)
    emb = (torch.randn(1, 20, 256, generator=torch.Generator().manual_seed(1)) * 0.5).to(torch.bfloat16).to(dev)
    a = llm.generate_from_embeds(emb, max_new_tokens=6)
    assert isinstance(next(iter(llm._decoders.values())), HipDecoder)
    b = llm.generate_from_embeds(emb, max_new_tokens=6, use_hip_decode=False)
    assert isinstance(next(iter(llm._decoders.values())), StaticDecoder) and not isinstance(next(iter(llm._decoders.values())), HipDecoder)
    assert a.shape == b.shape == (1, 6)
    emb2 = emb.repeat(2, 1, 1)  # batch 2: the PyTorch decoder
    llm.generate_from_embeds(emb2, max_new_tokens=3)
    assert not isinstance(next(iter(llm._decoders.values())), HipDecoder)


def test_gemv_forms_give_the_same_bits_across_processes(dev):
    """launch_decode_gemv picks between gemv_body and the x-in-LDS, whole-row-in-flight kernels by shape (MERV_GEMV_XLDS: probe hook read
    once per process): 15 launches -- plain, fused norm, gate-up pair, q / k / v, ragged rows and K -- must give the same bits with the
    form switched off ("0"), at its default and forced wherever it fits ("4")."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for mode in ("0", "", "4", "product"):
        env = dict(os.environ)
        env.pop("MERV_GEMV_XLDS", None)
        env.pop("MERV_HIP_LIB", None)
        env["MERV_TUNING_HOOKS"] = "1"  # the hooks build reads the variable (the product library reads none)
        if mode == "product":  # ... and the product library itself, with the variable set in its environment: ignored, same bits again
            env.pop("MERV_TUNING_HOOKS")
            env["MERV_GEMV_XLDS"] = "0"
        elif mode:
            env["MERV_GEMV_XLDS"] = mode
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "probes", "gemv_bits.py")], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 15, r.stdout
        outs.append(lines)
    assert outs[0] == outs[1] == outs[2] == outs[3]


def test_split_attention_with_weight_prefetch_blocks_gives_the_same_partials(dev):
    """merv_decode_attention_split_prefetch: the extra workgroups only READ the next launch's weights -- partials, cache rows and everything a generation
    returns are those of merv_decode_attention_split (opt-in path of HipDecoder: prefetch_oproj)."""
    from merv_amd.llm import HipDecoder, LlamaBackbone
    llm = LlamaBackbone(dict(vocab_size=320, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
                             num_key_value_heads=4, max_position_embeddings=2048, bos_token_id=1, eos_token_id=None, pad_token_id=0), device=dev)
    emb = (torch.randn(1, 37, 512, generator=torch.Generator().manual_seed(3)) * 0.5).to(torch.bfloat16).to(dev)
    was = HipDecoder.prefetch_oproj
    try:
        HipDecoder.prefetch_oproj = False
        ref = llm.generate_from_embeds(emb, max_new_tokens=24)
        llm._decoders.clear()
        HipDecoder.prefetch_oproj = True
        out = llm.generate_from_embeds(emb, max_new_tokens=24)
        dec = next(iter(llm._decoders.values()))
        assert isinstance(dec, HipDecoder) and dec.prefetch_oproj and (dec.H * dec.NSPLIT) % 8 == 0
    finally:
        HipDecoder.prefetch_oproj = was
        llm._decoders.clear()
    assert torch.equal(out, ref)
