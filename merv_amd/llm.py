"""
Thin LLM backbone for the hand-off at the end of the hot path: HF `LlamaForCausalLM` on
PyTorch-ROCm with SDPA attention (the reference hard-codes flash_attention_2 for training and SDPA for inference,
base_llm.py:121; no flash-attn / Triton dependency here), plus an explicit prefill(inputs_embeds) + KV-cache decode
loop instead of inheriting GenerationMixin through a non-PreTrainedModel class (base_vidlm.py:30,98-108 -- brittle
across transformers versions, SURVEY section 7 "hard parts").

No checkpoints can be fetched here: `LlamaBackbone(config)` builds the architecture with seeded random weights
(`llama2_7b_config()` = meta-llama/Llama-2-7b-hf geometry with the reference's <PAD> token / pad-to-64 resize,
llama2.py:74-76), or loads a local state dict.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib  # (binding only: the library itself is loaded when a HipDecoder is built)


def llama2_7b_config() -> Dict:
    return dict(vocab_size=32064, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                num_key_value_heads=32, max_position_embeddings=4096, rms_norm_eps=1e-5, bos_token_id=1, eos_token_id=2,
                pad_token_id=32000)


def llama2_13b_config() -> Dict:
    return dict(llama2_7b_config(), hidden_size=5120, intermediate_size=13824, num_hidden_layers=40, num_attention_heads=40,
                num_key_value_heads=40)


def mistral_7b_config() -> Dict:
    """mistralai/Mistral-7B-Instruct-v0.2 geometry (GQA 32/8, no sliding window) with the same <PAD> / pad-to-64 resize."""
    return dict(vocab_size=32064, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                num_key_value_heads=8, max_position_embeddings=32768, rms_norm_eps=1e-5, rope_theta=1e6, sliding_window=None,
                bos_token_id=1, eos_token_id=2, pad_token_id=32000)


def llama3_8b_config() -> Dict:
    """meta-llama/Meta-Llama-3-8B geometry (GQA 32/8, 128 256 tokens + the reference's <PAD>, resized to a multiple of 64:
    llama3.py:46-48)."""
    return dict(vocab_size=128320, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                num_key_value_heads=8, max_position_embeddings=8192, rms_norm_eps=1e-5, rope_theta=500000.0, bos_token_id=128000,
                eos_token_id=128001, pad_token_id=128256)


def llama31_8b_config() -> Dict:
    """meta-llama/Meta-Llama-3.1-8B-Instruct: the same tower with the "llama3" frequency-dependent rotary scaling and the
    tokenizer's own <|finetune_right_pad_id|> = 128004 as pad (llama3.py:101-102: no resize)."""
    return dict(llama3_8b_config(), vocab_size=128256, max_position_embeddings=131072, pad_token_id=128004,
                rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                              "original_max_position_embeddings": 8192})


def qwen25_7b_config() -> Dict:
    """Qwen/Qwen2.5-7B-Instruct geometry (GQA 28/4, q / k / v biases)."""
    return dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28,
                num_key_value_heads=4, max_position_embeddings=32768, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False,
                bos_token_id=151643, eos_token_id=151645, use_sliding_window=False)


def qwen25_3b_config() -> Dict:
    """Qwen/Qwen2.5-3B-Instruct geometry (GQA 16/2, tied embeddings)."""
    return dict(vocab_size=151936, hidden_size=2048, intermediate_size=11008, num_hidden_layers=36, num_attention_heads=16,
                num_key_value_heads=2, max_position_embeddings=32768, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=True,
                bos_token_id=151643, eos_token_id=151645, use_sliding_window=False)


class LlamaBackbone:
    """`family`: "llama" (LlamaForCausalLM: Llama-2 / Vicuna / Llama-3 / 3.1), "mistral" (MistralForCausalLM) or "qwen2"
    (Qwen2ForCausalLM); `identifier` is the reference's LLM registry key (materialize.py:76-101) and picks the prompt builder
    (llama2.py:78-89, llama3.py:50-56,104-105, qwen2.py:51-53)."""

    def __init__(self, config: Optional[Dict] = None, device="cuda:0", dtype=torch.bfloat16, seed: int = 0,
                 state_dict: Optional[Dict] = None, llm_max_length: int = 2048, family: str = "llama",
                 identifier: str = "llama2-7b-pure"):
        if family == "llama":
            from transformers import LlamaConfig as Cfg, LlamaForCausalLM as Cls
        elif family == "mistral":
            from transformers import MistralConfig as Cfg, MistralForCausalLM as Cls
        elif family == "qwen2":
            from transformers import Qwen2Config as Cfg, Qwen2ForCausalLM as Cls
        else:
            raise ValueError(f"unknown LLM family `{family}`")
        cfg = Cfg(**(config or llama2_7b_config()))
        cfg._attn_implementation = "sdpa"
        torch.manual_seed(seed)
        with torch.device(device):
            self.llm = Cls(cfg)
        self.llm = self.llm.to(dtype).eval().requires_grad_(False)
        self.device, self.dtype, self.llm_max_length = torch.device(device), dtype, llm_max_length
        self.embed_dim = cfg.hidden_size
        self.config = cfg
        self.identifier, self.family = identifier, family
        # merv.py:520-521 asks the TOKENIZER whether a BOS is prepended ("QWEN2.5-7B-INSTRUCT has no BOS token"); Qwen2.5's
        # config.json still carries bos_token_id = 151643, so the config is not the test. Used when no tokenizer object is
        # attached (vidlm.bos_token_length).
        self.prepends_bos = family != "qwen2"
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, state_dict: Dict, strict: bool = True):
        """Accepts the checkpoint's `llm_backbone` sub-dict (keys `llm.model.layers...`, merv.py:282: the reference's
        LLMBackbone holds the HF model as `self.llm`) or a bare HF state dict."""
        sd = {(k[4:] if k.startswith("llm.") else k): v for k, v in state_dict.items()}
        return self.llm.load_state_dict(sd, strict=strict)

    def state_dict(self) -> Dict:
        return {"llm." + k: v for k, v in self.llm.state_dict().items()}

    @property
    def prompt_builder_fn(self):
        from .prompting import (LLaMa2ChatPromptBuilder, LLaMa31PromptBuilder, MistralInstructPromptBuilder, PurePromptBuilder,
                                Qwen2PromptBuilder, VicunaV15ChatPromptBuilder)
        i = self.identifier
        if i.startswith("llama3.1-"):
            return LLaMa31PromptBuilder  # llama3.py:104-105
        if i.startswith("qwen2"):
            return Qwen2PromptBuilder  # qwen2.py:51-53
        if i.endswith("-pure"):
            return PurePromptBuilder
        if (i.startswith("llama2-") or i.startswith("llama3-")) and i.endswith("-chat"):
            return LLaMa2ChatPromptBuilder  # llama3.py:54-55 keeps the Llama-2 chat wrapper for llama3-*-chat
        if i.startswith("vicuna"):
            return VicunaV15ChatPromptBuilder
        if i.startswith("mistral") and i.endswith("-instruct"):
            return MistralInstructPromptBuilder
        raise ValueError(f"No PromptBuilder defined for LLM Backbone `{i}`")

    def embed_input_ids(self, input_ids: torch.LongTensor) -> torch.Tensor:
        return self.llm.get_input_embeddings()(input_ids)

    def _needs_hf_decode(self, total_len: int) -> bool:
        """True when the model has a feature neither static-cache decoder implements, so generation must stay on the HF
        module's own forward: sliding-window attention that would actually clip at this length (Mistral-v0.1-style
        `sliding_window`, Qwen2 with `use_sliding_window`), or biases on the MLP projections."""
        cfg = self.config
        win = getattr(cfg, "sliding_window", None)
        if win is not None and getattr(cfg, "use_sliding_window", True) and total_len > int(win):
            return True
        mlp = self.llm.model.layers[0].mlp
        return any(getattr(getattr(mlp, n, None), "bias", None) is not None for n in ("gate_proj", "up_proj", "down_proj"))

    @torch.inference_mode()
    def generate_from_embeds(self, inputs_embeds: torch.Tensor, max_new_tokens: int = 32, do_sample: bool = False,
                             temperature: float = 1.0, eos_token_id: Optional[int] = None,
                             generator: Optional[torch.Generator] = None, use_graph: bool = True, top_k: int = 0,
                             top_p: Optional[float] = 1.0, repetition_penalty: Optional[float] = 1.0, min_new_tokens: int = 0,
                             use_hip_decode: bool = True, prompt_ids: Optional[torch.LongTensor] = None) -> torch.LongTensor:
        """Prefill on `inputs_embeds` [B, S, D] (merv.py:723-734), then decode token by token on the KV cache
        (merv.py:524-538). Returns the new token ids [B, <= max_new_tokens]. Decoding controls are the subset of HF
        `generate` kwargs the reference's scripts pass through (merv.py:818-825): greedy or sampling with `temperature`,
        `top_k`, `top_p`, plus `repetition_penalty` (HF's logits processors, same order: penalty -> temperature -> top-k ->
        top-p). The reference hands `input_ids` to HF generate (merv.py:819), so HF's RepetitionPenaltyLogitsProcessor penalises
        the PROMPT's tokens as well as the generated ones: pass them as `prompt_ids` [B, S_text] (MERV.generate does) to get the
        same set; without them only generated tokens are penalised. `None` for top_p / repetition_penalty / top_k means "off",
        as in HF's GenerationConfig."""
        top_p = 1.0 if top_p is None else float(top_p)
        repetition_penalty = 1.0 if repetition_penalty is None else float(repetition_penalty)
        top_k = int(top_k or 0)
        dec = None
        if use_graph and self._needs_hf_decode(inputs_embeds.shape[1] + max_new_tokens + 1):
            use_graph = False  # the static-cache decoders implement neither sliding-window attention nor MLP biases
        if use_graph and inputs_embeds.is_cuda and inputs_embeds.shape[1] + max_new_tokens + 1 <= self.config.max_position_embeddings:
            # static-cache prefill + hipGraph-replayed decode steps (StaticDecoder above)
            need = inputs_embeds.shape[1] + max_new_tokens + 1
            bucket = min((need + 255) // 256 * 256, self.config.max_position_embeddings)  # cache length, reused across calls
            key = (bucket, inputs_embeds.shape[0], bool(use_hip_decode))
            if not hasattr(self, "_decoders"):
                self._decoders = {}
            if key not in self._decoders:
                self._decoders.clear()  # one static cache at a time (a 7B model's is ~0.5 GB per 1024 positions)
                cls = HipDecoder if (use_hip_decode and HipDecoder.supports(self.llm, inputs_embeds.shape[0])) else StaticDecoder
                self._decoders[key] = cls(self.llm, bucket, inputs_embeds.shape[0])
            dec = self._decoders[key]
            logits = dec.prefill(inputs_embeds)
            step = lambda tok: dec.decode(tok, use_graph=True)
        else:
            out = self.llm(inputs_embeds=inputs_embeds.to(self.dtype), use_cache=True)
            state = {"past": out.past_key_values}
            logits = out.logits[:, -1].float()

            def step(tok):
                o = self.llm(input_ids=tok[:, None], past_key_values=state["past"], use_cache=True)
                state["past"] = o.past_key_values
                return o.logits[:, -1].float()
        if (isinstance(dec, HipDecoder) and not do_sample and repetition_penalty == 1.0 and max_new_tokens > 0
                and (eos_token_id is None or min_new_tokens == 0) and dec.use_greedy_graph):
            # greedy search entirely on the device: every replayed step picks its own successor (HipDecoder.greedy_run)
            first = logits.argmax(-1)
            if max_new_tokens == 1 or (eos_token_id is not None and int(first) == eos_token_id):
                return first[None]  # nothing to decode: no step runs, no graph is captured
            rest = dec.greedy_run(first, max_new_tokens - 1, inputs_embeds.shape[1], eos_token_id)
            ids = torch.cat([first, rest])[None]
            if eos_token_id is not None:
                hit = (ids[0] == eos_token_id).nonzero()
                if hit.numel():
                    ids = ids[:, : int(hit[0]) + 1]
            return ids
        if (isinstance(dec, HipDecoder) and do_sample and top_k == 0 and top_p >= 1.0 and repetition_penalty == 1.0 and max_new_tokens > 0
                and dec.use_greedy_graph and hasattr(dec.lib, "merv_decode_sample_advance")):  # (an A/B library of an earlier round lacks the entry)
            # temperature sampling entirely on the device (HipDecoder.sample_run): the first token from the prompt's logits here, with the
            # caller's generator (which also seeds the device stream: one draw per generation, so a seeded generator reproduces the text)
            lg = logits / max(temperature, 1e-6)
            if eos_token_id is not None and min_new_tokens > 0:
                lg = lg.clone()
                lg[:, eos_token_id] = float("-inf")
            first = torch.multinomial(torch.softmax(lg, dim=-1), 1, generator=generator)[:, 0]
            seed = int(torch.randint(0, 2**62, (1,), generator=generator, device=generator.device if generator is not None else "cpu"))
            if max_new_tokens == 1 or (eos_token_id is not None and int(first) == eos_token_id):
                return first[None]
            rest = dec.sample_run(first, max_new_tokens - 1, inputs_embeds.shape[1], temperature, seed, eos_token_id, min_new_tokens)
            ids = torch.cat([first, rest])[None]
            if eos_token_id is not None:
                hit = (ids[0] == eos_token_id).nonzero()
                if hit.numel():
                    ids = ids[:, : int(hit[0]) + 1]
            return ids
        new_tokens = []
        done = torch.zeros(inputs_embeds.shape[0], dtype=torch.bool, device=inputs_embeds.device)
        for i in range(max_new_tokens):
            if eos_token_id is not None and i < min_new_tokens:  # HF MinLength / MinNewTokensLength processors
                logits = logits.clone()
                logits[:, eos_token_id] = float("-inf")
            if repetition_penalty != 1.0 and (new_tokens or prompt_ids is not None):  # HF RepetitionPenaltyLogitsProcessor
                seen = ([prompt_ids.to(logits.device)] if prompt_ids is not None else []) + ([torch.stack(new_tokens, 1)] if new_tokens else [])
                prev = torch.cat(seen, 1)
                sel = logits.gather(1, prev)
                logits = logits.scatter(1, prev, torch.where(sel < 0, sel * repetition_penalty, sel / repetition_penalty))
            if do_sample:
                lg = logits / max(temperature, 1e-6)
                if top_k and top_k > 0:  # HF TopKLogitsWarper
                    kth = lg.topk(min(top_k, lg.shape[-1]), dim=-1).values[:, -1:]
                    lg = lg.masked_fill(lg < kth, float("-inf"))
                if top_p < 1.0:  # HF TopPLogitsWarper: drop the low tail whose cumulative probability is <= 1 - top_p
                    srt, idx = lg.sort(dim=-1, descending=False)
                    drop = srt.softmax(-1).cumsum(-1) <= (1.0 - top_p)
                    drop[:, -1] = False
                    lg = lg.masked_fill(drop.scatter(1, idx, drop), float("-inf"))
                nxt = torch.multinomial(torch.softmax(lg, dim=-1), 1, generator=generator)[:, 0]
            else:
                nxt = logits.argmax(-1)
            new_tokens.append(nxt)
            if eos_token_id is not None:
                done |= nxt == eos_token_id
                if (i % 8 == 7 or i == max_new_tokens - 1) and bool(done.all()):  # one host sync per 8 tokens
                    break
            if i + 1 < max_new_tokens:
                logits = step(nxt)
        ids = torch.stack(new_tokens, 1)
        if eos_token_id is not None:  # the EOS test above runs every 8 tokens: cut at the step where every row had finished
            all_done = ((ids == eos_token_id).cumsum(1) > 0).all(0)
            if bool(all_done.any()):
                ids = ids[:, : int(all_done.float().argmax()) + 1]
        return ids


class StaticDecoder:
    """Prefill + token-by-token decode of a Llama / Mistral `*ForCausalLM` written with plain torch ops on a static KV
    cache, so that ONE decode step has fixed shapes and replays from a hipGraph (torch.cuda.CUDAGraph). HF's eager decode
    is launch-bound at batch 1 (≈350 small kernels per token); the graph removes the per-launch overhead. Still
    PyTorch-ROCm (library GEMMs, `F.scaled_dot_product_attention`): plumbing around the hand-off, not part of the HIP
    path. Uses the HF module's own parameters (no copies); rotary convention, RMSNorm and GQA as in
    transformers' modeling_llama / modeling_mistral (checked against the module's own forward in tests)."""

    def __init__(self, hf_model, max_len: int, batch: int = 1) -> None:
        import torch.nn.functional as F
        self.F = F
        self.m = hf_model
        cfg = hf_model.config
        self.cfg = cfg
        self.H, self.Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
        self.hd = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
        self.eps = cfg.rms_norm_eps
        self.max_len, self.B = max_len, batch
        p = next(hf_model.parameters())
        self.dev, self.dt = p.device, p.dtype
        rot = getattr(getattr(hf_model, "model", None), "rotary_emb", None)
        if rot is not None:
            # the module's own rotary embedding: plain RoPE (Llama-2, Mistral, Qwen2), Llama-3.1's frequency-dependent scaling, ...
            with torch.no_grad():
                cos, sin = rot(torch.zeros(1, dtype=self.dt, device=self.dev), torch.arange(max_len, device=self.dev)[None])
            self.cos, self.sin = cos[0].to(self.dt).contiguous(), sin[0].to(self.dt).contiguous()  # [max_len, hd]
        else:
            theta = getattr(cfg, "rope_theta", None)
            if theta is None and isinstance(getattr(cfg, "rope_parameters", None), dict):
                theta = cfg.rope_parameters.get("rope_theta")
            theta = float(theta or 10000.0)
            inv = 1.0 / (theta ** (torch.arange(0, self.hd, 2, dtype=torch.float32, device=self.dev) / self.hd))
            fr = torch.outer(torch.arange(max_len, dtype=torch.float32, device=self.dev), inv)
            emb = torch.cat([fr, fr], dim=-1)
            self.cos, self.sin = emb.cos().to(self.dt), emb.sin().to(self.dt)  # [max_len, hd]
        L = cfg.num_hidden_layers
        self.K = [torch.zeros(batch, self.Hkv, max_len, self.hd, dtype=self.dt, device=self.dev) for _ in range(L)]
        self.V = [torch.zeros(batch, self.Hkv, max_len, self.hd, dtype=self.dt, device=self.dev) for _ in range(L)]
        self.ar = torch.arange(max_len, device=self.dev)
        # the token a step embeds: ONE tensor for the decoder's life. Every captured graph (decode(), HipDecoder.greedy_run) reads and
        # writes THIS storage; callers only ever copy_ into it (rebinding it after a capture would leave that graph on freed memory).
        self.tok = torch.zeros(batch, 1, dtype=torch.long, device=self.dev)
        self.graph = None

    def _rms(self, x, w):
        xf = x.float()
        return w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)).to(x.dtype)

    @staticmethod
    def _rot(x):
        h = x.shape[-1] // 2
        return torch.cat([-x[..., h:], x[..., :h]], dim=-1)

    def _layer(self, li, x, cos, sin, pos_idx, mask, S):
        F = self.F
        lyr = self.m.model.layers[li]
        a = lyr.self_attn
        B = x.shape[0]
        h = self._rms(x, lyr.input_layernorm.weight)
        q = F.linear(h, a.q_proj.weight, a.q_proj.bias).view(B, S, self.H, self.hd).transpose(1, 2)
        k = F.linear(h, a.k_proj.weight, a.k_proj.bias).view(B, S, self.Hkv, self.hd).transpose(1, 2)
        v = F.linear(h, a.v_proj.weight, a.v_proj.bias).view(B, S, self.Hkv, self.hd).transpose(1, 2)
        q = q * cos + self._rot(q) * sin
        k = k * cos + self._rot(k) * sin
        self.K[li].index_copy_(2, pos_idx, k)
        self.V[li].index_copy_(2, pos_idx, v)
        if mask is None:  # prefill: causal over the S new positions only (the cache beyond them is not read)
            o = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=self.H != self.Hkv)
        else:  # decode: one query against the whole static cache, slots > position masked off
            o = F.scaled_dot_product_attention(q, self.K[li], self.V[li], attn_mask=mask, enable_gqa=self.H != self.Hkv)
        x = x + F.linear(o.transpose(1, 2).reshape(B, S, self.H * self.hd), a.o_proj.weight, a.o_proj.bias)
        h = self._rms(x, lyr.post_attention_layernorm.weight)
        mlp = lyr.mlp
        return x + F.linear(F.silu(F.linear(h, mlp.gate_proj.weight)) * F.linear(h, mlp.up_proj.weight), mlp.down_proj.weight)

    @torch.inference_mode()
    def prefill(self, inputs_embeds: torch.Tensor) -> torch.Tensor:
        """inputs_embeds [B, S, D] -> logits of the last position [B, vocab] (fp32); fills cache slots 0..S-1."""
        B, S, _ = inputs_embeds.shape
        assert B == self.B and S < self.max_len
        x = inputs_embeds.to(self.dt)
        pos = self.ar[:S]
        cos, sin = self.cos[:S][None, None], self.sin[:S][None, None]
        for li in range(self.cfg.num_hidden_layers):
            x = self._layer(li, x, cos, sin, pos, None, S)
        if not hasattr(self, "pos"):
            self.pos = torch.tensor([S], device=self.dev)
        else:
            self.pos.fill_(S)  # same tensor: a captured decode graph keeps reading it
        return self.F.linear(self._rms(x[:, -1:], self.m.model.norm.weight), self.m.lm_head.weight)[:, 0].float()

    def _step(self):
        x = self.m.model.embed_tokens(self.tok)
        cos = self.cos.index_select(0, self.pos)[None, None]
        sin = self.sin.index_select(0, self.pos)[None, None]
        mask = (self.ar <= self.pos)[None, None, None, :]
        for li in range(self.cfg.num_hidden_layers):
            x = self._layer(li, x, cos, sin, self.pos, mask, 1)
        return self.F.linear(self._rms(x, self.m.model.norm.weight), self.m.lm_head.weight)[:, 0].float()

    @torch.inference_mode()
    def decode(self, token: torch.Tensor, use_graph: bool = True) -> torch.Tensor:
        """token [B] (the token at position self.pos) -> logits for the next position [B, vocab] (fp32)."""
        self.tok.copy_(token[:, None])
        if self.graph is None:
            if not use_graph:
                out = self._step()
                self.pos += 1
                return out
            self._step()  # warm-up outside the capture (same slot is rewritten identically by the replay)
            torch.cuda.synchronize(self.dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.logits = self._step()
        if use_graph:
            self.graph.replay()
            self.pos += 1
            return self.logits
        out = self._step()
        self.pos += 1
        return out


class HipDecoder(StaticDecoder):
    """The batch-1 decode step of `StaticDecoder` on libmerv_hip.so's decode kernels (csrc/decode.hip): 5 launches per layer
    -- the q / k / v projections as one GEMV launch with the input RMSNorm fused in, rotary + cache update + split attention + merge as one launch, o-projection with the residual, the
    gated MLP as one GEMV pair with its RMSNorm and silu * up fused, down-projection with the residual -- instead of ~35 PyTorch ones, each a
    pure HBM stream (weights read once, non-temporal). Same parameters (the HF module's, no copies), same static cache and
    rotary tables, same rounding points as the bf16 module; the prefill keeps its GEMMs on PyTorch-ROCm and takes the rest from the
    library too (`_prefill_fused`). Measured on MI355X
    with Llama-2-7B geometry: see DESIGN.md section 6 (e2e)."""

    NSPLIT = 8  # position ranges per head in decode attention: 32 heads x 8 = one block per CU

    @staticmethod
    def supports(hf_model, batch: int) -> bool:
        cfg = hf_model.config
        hd = getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads
        p = next(hf_model.parameters())
        lyr = hf_model.model.layers[0]
        mlp_bias = any(getattr(getattr(lyr.mlp, n, None), "bias", None) is not None for n in ("gate_proj", "up_proj", "down_proj"))
        return (batch == 1 and hd == 128 and p.is_cuda and p.dtype == torch.bfloat16 and lyr.self_attn.o_proj.bias is None
                and not mlp_bias and cfg.hidden_size % 8 == 0 and cfg.intermediate_size % 8 == 0)

    def __init__(self, hf_model, max_len: int, batch: int = 1) -> None:
        super().__init__(hf_model, max_len, batch)
        from . import _lib
        self.lib = _lib.load()
        cfg = self.cfg
        D, I = cfg.hidden_size, cfg.intermediate_size
        mk = lambda n, dt=self.dt: torch.empty(n, dtype=dt, device=self.dev)
        self.x, self.h, self.ao, self.mid = mk(D), mk(D), mk(self.H * self.hd), mk(I)
        self.q, self.k, self.v = mk(self.H * self.hd), mk(self.Hkv * self.hd), mk(self.Hkv * self.hd)
        # partials of the split attention + one arrival counter per head (zero between launches: the kernel restores them)
        self.ws = torch.zeros(max(self.lib.merv_decode_attention_fused_workspace_floats(self.H, self.NSPLIT),
                                  self.lib.merv_decode_attention_split_workspace_floats(self.H, self.NSPLIT)), dtype=torch.float32, device=self.dev)
        self.logits32 = torch.empty(1, cfg.vocab_size, dtype=torch.float32, device=self.dev)
        # greedy_run's token log (token chosen AT position p, for p + 1): one tensor for the decoder's life, like self.tok
        self.out_tokens = torch.zeros(max_len, dtype=torch.long, device=self.dev)
        if self.use_hip_prefill and self.fuse_qkv:
            for lyr in hf_model.model.layers:  # re-homes q / k / v of every layer once, here (see _qkv_fused)
                self._qkv_fused(lyr.self_attn)

    # the attention launch ends at its split partials and the o-projection merges them under its first weight trip (bit-identical to
    # the fused attention launch + the plain o-projection; 4.8 us per layer faster): default; MERV_DECODE_SPLIT_MERGE=0 restores them.
    # (Round 4 also built attention + o-projection and the whole step as ONE launch; both bit-identical and slower, removed in round 5:
    # git 05f38ce / 0dad582, EXPERIMENTS.md section 5.)
    use_split_merge = _lib.tuning("MERV_DECODE_SPLIT_MERGE", "1") != "0"
    # round 6, measured and NOT the default: the split attention launch also touching the o-projection's weights, each slice on the XCD whose L2 its
    # reader uses (merv_decode_attention_split_prefetch). The o-projection gains 2.2 us per layer, the attention launch loses 3.4 (33.5 MB are 5.6 us of
    # HBM stream: more than the launch's idle time): 2.60-2.64 -> 2.63 ms per token (profiles/r06_decode_mall_bound.json). MERV_DECODE_PREFETCH_WO=1 opts in.
    prefetch_oproj = _lib.tuning("MERV_DECODE_PREFETCH_WO", "0") == "1"
    # greedy generation with the argmax / token hand-over / position increment inside the captured step; MERV_DECODE_GREEDY_GRAPH=0: host loop
    use_greedy_graph = _lib.tuning("MERV_DECODE_GREEDY_GRAPH", "1") != "0"

    def prefill(self, inputs_embeds: torch.Tensor) -> torch.Tensor:
        # the fused attention launch expects its per-head arrival counters at zero and restores them itself; an aborted launch
        # (a fault, a killed process sharing the buffer) would leave them non-zero and every later merge would misfire -- so every
        # generation starts from a zeroed workspace (one memset per generate(), nothing per token)
        self.ws.zero_()
        if not self.use_hip_prefill:
            return super().prefill(inputs_embeds)
        return self._prefill_fused(inputs_embeds)

    # Prefill with the elementwise parts of every layer on libmerv_hip.so (RMSNorm, rotary + cache fill, silu * up: one launch each
    # instead of ~35 PyTorch kernels per layer; same rounding points as `_layer`), the GEMMs and the causal attention on PyTorch-ROCm.
    # MERV_HIP_PREFILL=0 takes the plain PyTorch expression of StaticDecoder.
    use_hip_prefill = _lib.tuning("MERV_HIP_PREFILL", "1") != "0"
    # ... and the causal attention on merv_prefill_attention (head dim 128) instead of PyTorch-ROCm's SDPA; MERV_HIP_PREFILL_ATTN=0: SDPA
    use_hip_prefill_attn = _lib.tuning("MERV_HIP_PREFILL_ATTN", "1") != "0"

    # q / k / v of the prompt as one library GEMM (below); MERV_PREFILL_FUSE_QKV=0 keeps the module's three parameters untouched
    fuse_qkv = _lib.tuning("MERV_PREFILL_FUSE_QKV", "1") != "0"

    def _qkv_fused(self, attn):
        """The q / k / v projection weights (and biases) of one attention module as ONE [Nq + Nk + Nv, D] matrix, so the prompt's
        three projections are one library GEMM (105 us against 3 x 45 at 1049 tokens). No second copy: the three parameters are
        re-homed as row ranges of the fused tensor (`param.data = fused[a:b]`, same values, contiguous rows -- what the decode
        kernels read through their own pointers), so in-place loads keep updating it.

        THIS MUTATES THE HF MODULE, once, when the decoder is built (not as a side effect of a later generate()): afterwards the
        three parameters of a layer share one storage. `state_dict()` / `load_state_dict()` / in-place optimiser updates are
        unaffected (each parameter is still its own contiguous row range); what notices is code that checks for shared storage --
        `safetensors.torch.save_file` / `save_pretrained(safe_serialization=True)` refuse aliasing tensors: clone the state dict
        first, or build the decoder with MERV_PREFILL_FUSE_QKV=0 ("LLM hand-off"). A parameter whose storage was
        REPLACED since (`param.data = ...`, `load_state_dict(assign=True)`) is detected by its address at the next prefill: the layer
        is fused again and every captured decode graph is dropped, because the graphs hold the old weight pointers."""
        if not self.fuse_qkv:
            return None
        projs = (attn.q_proj, attn.k_proj, attn.v_proj)
        fused = getattr(attn, "_merv_qkv", None)
        ok = fused is not None
        if ok:
            off = 0
            for pr in projs:
                n = pr.weight.shape[0]
                ok = ok and pr.weight.data_ptr() == fused[0][off:off + n].data_ptr() and pr.weight.is_contiguous()
                if pr.bias is not None:
                    ok = ok and fused[1] is not None and pr.bias.data_ptr() == fused[1][off:off + n].data_ptr()
                off += n
        if not ok:
            has_b = all(pr.bias is not None for pr in projs)
            if not has_b and any(pr.bias is not None for pr in projs):
                return None  # mixed biases: no HF family does this; keep the three GEMMs
            with torch.inference_mode(False), torch.no_grad():  # ordinary tensors: the parameters stay usable under autograd
                w = torch.cat([pr.weight.data for pr in projs], 0)
                b = torch.cat([pr.bias.data for pr in projs], 0) if has_b else None
            off = 0
            for pr in projs:
                n = pr.weight.shape[0]
                pr.weight.data = w[off:off + n]
                if has_b:
                    pr.bias.data = b[off:off + n]
                off += n
            if fused is not None:  # a re-fuse: the captured steps read the storage the parameters had before
                self.graph = None
                self.greedy_graph = self.greedy_graph_chunk = None
                self.sample_graph = self.sample_graph_chunk = None
            fused = attn._merv_qkv = (w, b)
        return fused

    @torch.inference_mode()
    def _prefill_fused(self, inputs_embeds: torch.Tensor) -> torch.Tensor:
        from ._lib import check, ptr
        F, lib, m = self.F, self.lib, self.m
        B, S, D = inputs_embeds.shape
        assert B == 1 and S < self.max_len
        H, Hkv, hd, I = self.H, self.Hkv, self.hd, self.cfg.intermediate_size
        with torch.cuda.device(self.dev):
            st = torch.cuda.current_stream(self.dev).cuda_stream
            x = inputs_embeds[0].to(self.dt).contiguous()
            if x.data_ptr() == inputs_embeds.data_ptr():
                x = x.clone()  # the residual stream is updated in place
            h = torch.empty_like(x)

            def rms(src, w, dst, rows):
                check(lib.merv_decode_rmsnorm(ptr(src), ptr(w), ptr(dst), rows, D, self.eps, st), "merv_decode_rmsnorm")

            def add_rms(delta, w):  # x += delta; h = RMSNorm(x) * w, one pass
                check(lib.merv_add_rmsnorm(ptr(x), ptr(delta), ptr(w), ptr(h), S, D, self.eps, st), "merv_add_rmsnorm")

            fuse_add = D <= 8192
            layers = m.model.layers
            rms(x, layers[0].input_layernorm.weight, h, S)
            for li, lyr in enumerate(layers):
                a, mlp = lyr.self_attn, lyr.mlp
                fq = self._qkv_fused(a)
                if fq is not None:  # one GEMM, q / k / v are column ranges of its output
                    qkv = F.linear(h, fq[0], fq[1])
                    ld = qkv.shape[1]
                    qp, kp, vp, ldq, ldk = qkv.data_ptr(), qkv.data_ptr() + 2 * H * hd, qkv.data_ptr() + 2 * (H + Hkv) * hd, ld, ld
                    q = qkv[:, :H * hd]
                else:
                    q = F.linear(h, a.q_proj.weight, a.q_proj.bias)
                    k = F.linear(h, a.k_proj.weight, a.k_proj.bias)
                    v = F.linear(h, a.v_proj.weight, a.v_proj.bias)
                    qp, kp, vp, ldq, ldk = ptr(q), ptr(k), ptr(v), H * hd, Hkv * hd
                check(lib.merv_prefill_rope_cache(qp, kp, vp, ptr(self.K[li]), ptr(self.V[li]), ptr(self.cos), ptr(self.sin),
                                                  S, 0, H, Hkv, hd, self.max_len, ldq, ldk, st), "merv_prefill_rope_cache")
                if self.use_hip_prefill_attn:  # causal attention straight off the cache rows, output in the o-projection's layout
                    o = torch.empty(S, H * hd, dtype=self.dt, device=self.dev)
                    check(lib.merv_prefill_attention(qp, ptr(self.K[li]), ptr(self.V[li]), ptr(o), S, H, Hkv, hd, ldq, hd,
                                                     self.max_len * hd, H * hd, hd**-0.5, st), "merv_prefill_attention")
                else:
                    o = F.scaled_dot_product_attention(q.reshape(1, S, H, hd).transpose(1, 2), self.K[li][:, :, :S], self.V[li][:, :, :S],
                                                       is_causal=True, enable_gqa=H != Hkv).transpose(1, 2).reshape(S, H * hd)
                o = F.linear(o, a.o_proj.weight, a.o_proj.bias)
                if fuse_add:
                    add_rms(o, lyr.post_attention_layernorm.weight)
                else:
                    x += o
                    rms(x, lyr.post_attention_layernorm.weight, h, S)
                g = F.linear(h, mlp.gate_proj.weight)
                u = F.linear(h, mlp.up_proj.weight)
                check(lib.merv_silu_mul(ptr(g), ptr(u), ptr(g), S * I, st), "merv_silu_mul")
                dn = F.linear(g, mlp.down_proj.weight)
                if li + 1 == len(layers):
                    x += dn  # only the last row goes through the final norm below
                elif fuse_add:
                    add_rms(dn, layers[li + 1].input_layernorm.weight)
                else:
                    x += dn
                    rms(x, layers[li + 1].input_layernorm.weight, h, S)
            if not hasattr(self, "pos"):
                self.pos = torch.tensor([S], device=self.dev)
            else:
                self.pos.fill_(S)  # same tensor: a captured decode graph keeps reading it
            last = torch.empty(1, D, dtype=self.dt, device=self.dev)
            rms(x[S - 1:], m.model.norm.weight, last, 1)
            return F.linear(last, m.lm_head.weight).float()

    # Greedy decoding with nothing but the graph replay per token: the captured step ends with merv_decode_greedy_advance (argmax ->
    # next token, token log, position + 1, all on the device) instead of the host loop's argmax / copy / add kernels (23 us per token).
    greedy_graph = greedy_graph_chunk = None
    GREEDY_CHUNK = 8

    @torch.inference_mode()
    def greedy_run(self, token: torch.Tensor, steps: int, start: int, eos_token_id: Optional[int] = None) -> torch.Tensor:
        """token [1]: the token at position `start` (= self.pos). Runs up to `steps` decode steps, each choosing its successor by
        argmax; returns their tokens [n] (int64, device), n <= steps: with an `eos_token_id` the host looks every 8 steps and stops
        after the chunk that produced it."""
        from ._lib import check, ptr
        if steps <= 0:
            return torch.empty(0, dtype=torch.long, device=self.dev)
        self.tok.copy_(token[:, None])  # (never rebound: decode()'s graph and these graphs embed from the same tensor)
        if self.greedy_graph is None:
            self._step()  # warm-up outside the capture; the advance is not run: it would move the position
            torch.cuda.synchronize(self.dev)

            def capture(nsteps):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(nsteps):
                        self._step()
                        check(self.lib.merv_decode_greedy_advance(ptr(self.logits32), self.cfg.vocab_size, ptr(self.tok), ptr(self.pos),
                                                                  ptr(self.out_tokens), 0, torch.cuda.current_stream(self.dev).cuda_stream),
                              "merv_decode_greedy_advance")
                return g
            # GREEDY_CHUNK steps per replay as well as one: between two replays the GPU idles ~25 us (measured: the host loop's
            # three small kernels per token had been hiding in that gap), a chunk pays it once
            self.greedy_graph = capture(1)
            self.greedy_graph_chunk = capture(self.GREEDY_CHUNK)
        done = 0
        while done < steps:
            if steps - done >= self.GREEDY_CHUNK:
                self.greedy_graph_chunk.replay()
                done += self.GREEDY_CHUNK
            else:
                self.greedy_graph.replay()
                done += 1
            if eos_token_id is not None and (done % self.GREEDY_CHUNK == 0 or done == steps):  # one host look per chunk
                if bool((self.out_tokens[start:start + done] == eos_token_id).any()):
                    break
        return self.out_tokens[start:start + done].clone()

    # Temperature sampling with nothing but the graph replay per token (round 6): as greedy_run, the captured step ending in
    # merv_decode_sample_advance -- Gumbel-max on a Philox stream keyed by (seed, position): exactly softmax(logits / T)'s categorical
    # distribution -- instead of the host loop's divide / softmax / multinomial / copy kernels and one host round trip per token.
    # Temperature, seed and the end-of-sequence rule are read from device memory (self.sample_params), so one pair of graphs serves
    # every generation. top-k / top-p / repetition penalty keep the host loop (generate_from_embeds).
    sample_graph = sample_graph_chunk = None

    @torch.inference_mode()
    def sample_run(self, token: torch.Tensor, steps: int, start: int, temperature: float, seed: int, eos_token_id: Optional[int] = None,
                   min_new_tokens: int = 0) -> torch.Tensor:
        """token [1]: the token at position `start` (= self.pos; new token number 0). Runs up to `steps` decode steps, each DRAWING its
        successor from softmax(logits / temperature); returns their tokens [n] (int64, device), n <= steps (with an `eos_token_id` the
        host looks every 8 steps and stops after the chunk that produced it; the token is barred while fewer than `min_new_tokens`
        have been drawn)."""
        import struct
        from ._lib import check, ptr
        if steps <= 0:
            return torch.empty(0, dtype=torch.long, device=self.dev)
        self.tok.copy_(token[:, None])
        blob = struct.pack("<fiQqq", 1.0 / max(float(temperature), 1e-6), -1 if eos_token_id is None else int(eos_token_id),
                           int(seed) & (2**64 - 1), int(min_new_tokens), int(start))
        if not hasattr(self, "sample_params"):
            self.sample_params = torch.zeros(32, dtype=torch.uint8, device=self.dev)
        self.sample_params.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
        if self.sample_graph is None:
            self._step()  # warm-up outside the capture; the advance is not run: it would move the position
            torch.cuda.synchronize(self.dev)

            def capture(nsteps):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(nsteps):
                        self._step()
                        check(self.lib.merv_decode_sample_advance(ptr(self.logits32), self.cfg.vocab_size, ptr(self.sample_params), ptr(self.tok),
                                                                  ptr(self.pos), ptr(self.out_tokens), 0,
                                                                  torch.cuda.current_stream(self.dev).cuda_stream), "merv_decode_sample_advance")
                return g
            self.sample_graph = capture(1)
            self.sample_graph_chunk = capture(self.GREEDY_CHUNK)
        done = 0
        while done < steps:
            if steps - done >= self.GREEDY_CHUNK:
                self.sample_graph_chunk.replay()
                done += self.GREEDY_CHUNK
            else:
                self.sample_graph.replay()
                done += 1
            if eos_token_id is not None and (done % self.GREEDY_CHUNK == 0 or done == steps):  # one host look per chunk
                if bool((self.out_tokens[start:start + done] == eos_token_id).any()):
                    break
        return self.out_tokens[start:start + done].clone()

    def _step(self):
        from ._lib import check, ptr
        lib, m = self.lib, self.m
        with torch.cuda.device(self.dev):
            st = torch.cuda.current_stream(self.dev).cuda_stream
            D, I, H, Hkv, hd = self.cfg.hidden_size, self.cfg.intermediate_size, self.H, self.Hkv, self.hd
            self.x.copy_(m.model.embed_tokens(self.tok).reshape(-1))
            x, h, pos = ptr(self.x), ptr(self.h), ptr(self.pos)

            def gemv(W, W2, xin, res, y, N, K, y32=0, norm=None):
                check(lib.merv_decode_gemv(ptr(W), 0 if W2 is None else ptr(W2), xin, res, y, y32, N, K,
                                           0 if norm is None else ptr(norm), self.eps, st), "merv_decode_gemv")

            for li, lyr in enumerate(m.model.layers):
                a, mlp = lyr.self_attn, lyr.mlp
                # input_layernorm fused into the q / k / v launch, post_attention_layernorm into the gate / up launch
                bq, bk, bv = a.q_proj.bias, a.k_proj.bias, a.v_proj.bias  # Qwen2 has them, Llama / Mistral do not
                check(lib.merv_decode_gemv3_bias(ptr(a.q_proj.weight), ptr(a.k_proj.weight), ptr(a.v_proj.weight), x, ptr(self.q), ptr(self.k),
                                                 ptr(self.v), H * hd, Hkv * hd, Hkv * hd, D, ptr(lyr.input_layernorm.weight), self.eps,
                                                 0 if bq is None else ptr(bq), 0 if bk is None else ptr(bk), 0 if bv is None else ptr(bv), st),
                      "merv_decode_gemv3_bias")
                if self.use_split_merge and H <= 256:
                    # rotary + cache + split attention; then x += o_proj(merge of the splits): the merge rides under the weight loads
                    if self.prefetch_oproj and (H * self.NSPLIT) % 8 == 0 and D % 16 == 0:
                        # ... with W_o touched by extra workgroups meanwhile, each on the XCD whose L2 the o-projection's workgroup of the same
                        # index reads through (16 rows per workgroup there): the launch leaves HBM idle, the next one then starts from L2
                        wo = a.o_proj.weight
                        check(lib.merv_decode_attention_split_prefetch(ptr(self.q), ptr(self.k), ptr(self.v), ptr(self.cos), ptr(self.sin), pos,
                                                                       ptr(self.K[li]), ptr(self.V[li]), ptr(self.ws), H, Hkv, hd, self.max_len,
                                                                       self.NSPLIT, hd**-0.5, ptr(wo), wo.numel() * 2, 16 * wo.shape[1] * 2, st),
                              "merv_decode_attention_split_prefetch")
                    else:
                        check(lib.merv_decode_attention_split(ptr(self.q), ptr(self.k), ptr(self.v), ptr(self.cos), ptr(self.sin), pos, ptr(self.K[li]),
                                                              ptr(self.V[li]), ptr(self.ws), H, Hkv, hd, self.max_len, self.NSPLIT, hd**-0.5, st),
                              "merv_decode_attention_split")
                    check(lib.merv_decode_oproj_merge(ptr(a.o_proj.weight), x, x, ptr(self.ws), 0, D, H, hd, self.NSPLIT, st),
                          "merv_decode_oproj_merge")
                else:
                    check(lib.merv_decode_attention_fused(ptr(self.q), ptr(self.k), ptr(self.v), ptr(self.cos), ptr(self.sin), pos, ptr(self.K[li]),
                                                          ptr(self.V[li]), ptr(self.ao), ptr(self.ws), H, Hkv, hd, self.max_len, self.NSPLIT,
                                                          hd**-0.5, st), "merv_decode_attention_fused")
                    gemv(a.o_proj.weight, None, ptr(self.ao), x, x, D, H * hd)  # x += o_proj(attn)
                gemv(mlp.gate_proj.weight, mlp.up_proj.weight, x, 0, ptr(self.mid), I, D, norm=lyr.post_attention_layernorm.weight)
                gemv(mlp.down_proj.weight, None, ptr(self.mid), x, x, D, I)  # x += down_proj(...)
            gemv(m.lm_head.weight, None, x, 0, 0, self.cfg.vocab_size, D, y32=ptr(self.logits32), norm=m.model.norm.weight)
        return self.logits32


# === Language Model Registry (materialize.py:76-101, llama2.py:24-51): ids wired on this path -> (family, geometry) ===
LLM_BACKBONES = {
    "llama2-7b-pure": ("llama", llama2_7b_config), "llama2-7b-chat": ("llama", llama2_7b_config),
    "llama2-13b-pure": ("llama", llama2_13b_config), "llama2-13b-chat": ("llama", llama2_13b_config),
    "vicuna-v15-7b": ("llama", llama2_7b_config), "vicuna-v15-13b": ("llama", llama2_13b_config),
    # BASELINE.json configs[4]: the Mistral-7B-Instruct swap, a new backbone modelled on llama2.py:55-98
    "mistral-v0.2-7b-pure": ("mistral", mistral_7b_config), "mistral-v0.2-7b-instruct": ("mistral", mistral_7b_config),
    # the reference's remaining registry keys (materialize.py:90-100)
    "llama3-8b-pure": ("llama", llama3_8b_config), "llama3-8b-chat": ("llama", llama3_8b_config),
    "llama3.1-8b-chat": ("llama", llama31_8b_config),
    "qwen2.5-7b-instruct": ("qwen2", qwen25_7b_config), "qwen2.5-3b-instruct": ("qwen2", qwen25_3b_config),
}


def get_llm_backbone_and_tokenizer(llm_backbone_id: str, llm_max_length: int = 2048, hf_token: Optional[str] = None,
                                   inference_mode: bool = False, *, config: Optional[Dict] = None,
                                   state_dict: Optional[Dict] = None, tokenizer=None, tokenizer_path: Optional[str] = None,
                                   device="cuda:0", seed: int = 0):
    """materialize.py:132-149. The hub is unreachable here, so weights come from `state_dict` (else seeded random) and the
    tokenizer from `tokenizer` / a local `tokenizer_path` (else None: generate() then takes and returns token ids).
    `config` overrides the registry geometry (reduced-size test models)."""
    if llm_backbone_id not in LLM_BACKBONES:
        raise ValueError(f"LLM Backbone `{llm_backbone_id}` is not supported!")
    family, geom = LLM_BACKBONES[llm_backbone_id]
    llm = LlamaBackbone(config or geom(), device=device, seed=seed, state_dict=state_dict, llm_max_length=llm_max_length,
                        family=family, identifier=llm_backbone_id)
    if tokenizer is None and tokenizer_path is not None:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(tokenizer_path, model_max_length=llm_max_length, local_files_only=True)
        if tok.pad_token is None:
            tok.add_special_tokens({"pad_token": "<PAD>"})  # llama2.py:74
        tokenizer = HFTokenizerAdapter(tok)
    return llm, tokenizer


class HFTokenizerAdapter:
    """`tokenizer(text) -> list[int]` (BOS included, as `tokenizer(prompt, return_tensors="pt").input_ids`,
    merv.py:789) and `decode(ids)` with special tokens skipped (merv.py:827)."""

    def __init__(self, hf_tokenizer) -> None:
        self.tok = hf_tokenizer

    def __call__(self, text: str):
        return list(self.tok(text, truncation=True).input_ids)

    def decode(self, ids) -> str:
        return self.tok.decode(ids, skip_special_tokens=True)
