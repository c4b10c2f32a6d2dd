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


class LlamaBackbone:
    """`family`: "llama" (LlamaForCausalLM) or "mistral" (MistralForCausalLM); `identifier` is the reference's LLM
    registry key (materialize.py:76-101) and picks the prompt builder (llama2.py:78-89)."""

    def __init__(self, config: Optional[Dict] = None, device="cuda:0", dtype=torch.bfloat16, seed: int = 0,
                 state_dict: Optional[Dict] = None, llm_max_length: int = 2048, family: str = "llama",
                 identifier: str = "llama2-7b-pure"):
        if family == "llama":
            from transformers import LlamaConfig as Cfg, LlamaForCausalLM as Cls
        elif family == "mistral":
            from transformers import MistralConfig as Cfg, MistralForCausalLM as Cls
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
        from .prompting import LLaMa2ChatPromptBuilder, MistralInstructPromptBuilder, PurePromptBuilder, VicunaV15ChatPromptBuilder
        i = self.identifier
        if i.endswith("-pure"):
            return PurePromptBuilder
        if i.startswith("llama2-") and i.endswith("-chat"):
            return LLaMa2ChatPromptBuilder
        if i.startswith("vicuna"):
            return VicunaV15ChatPromptBuilder
        if i.startswith("mistral") and i.endswith("-instruct"):
            return MistralInstructPromptBuilder
        raise ValueError(f"No PromptBuilder defined for LLM Backbone `{i}`")

    def embed_input_ids(self, input_ids: torch.LongTensor) -> torch.Tensor:
        return self.llm.get_input_embeddings()(input_ids)

    @torch.inference_mode()
    def generate_from_embeds(self, inputs_embeds: torch.Tensor, max_new_tokens: int = 32, do_sample: bool = False,
                             temperature: float = 1.0, eos_token_id: Optional[int] = None,
                             generator: Optional[torch.Generator] = None) -> torch.LongTensor:
        """Prefill on `inputs_embeds` [B, S, D] (merv.py:723-734), then decode token by token on the KV cache
        (merv.py:524-538). Returns the new token ids [B, <= max_new_tokens]."""
        out = self.llm(inputs_embeds=inputs_embeds.to(self.dtype), use_cache=True)
        past = out.past_key_values
        logits = out.logits[:, -1].float()
        new_tokens = []
        done = torch.zeros(inputs_embeds.shape[0], dtype=torch.bool, device=inputs_embeds.device)
        for _ in range(max_new_tokens):
            if do_sample:
                probs = torch.softmax(logits / max(temperature, 1e-6), dim=-1)
                nxt = torch.multinomial(probs, 1, generator=generator)[:, 0]
            else:
                nxt = logits.argmax(-1)
            new_tokens.append(nxt)
            if eos_token_id is not None:
                done |= nxt == eos_token_id
                if bool(done.all()):
                    break
            out = self.llm(input_ids=nxt[:, None], past_key_values=past, use_cache=True)
            past = out.past_key_values
            logits = out.logits[:, -1].float()
        return torch.stack(new_tokens, 1)


# === Language Model Registry (materialize.py:76-101, llama2.py:24-51): ids wired on this path -> (family, geometry) ===
LLM_BACKBONES = {
    "llama2-7b-pure": ("llama", llama2_7b_config), "llama2-7b-chat": ("llama", llama2_7b_config),
    "llama2-13b-pure": ("llama", llama2_13b_config), "llama2-13b-chat": ("llama", llama2_13b_config),
    "vicuna-v15-7b": ("llama", llama2_7b_config), "vicuna-v15-13b": ("llama", llama2_13b_config),
    # BASELINE.json configs[4]: the Mistral-7B-Instruct swap, a new backbone modelled on llama2.py:55-98
    "mistral-v0.2-7b-pure": ("mistral", mistral_7b_config), "mistral-v0.2-7b-instruct": ("mistral", mistral_7b_config),
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
