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


class LlamaBackbone:
    def __init__(self, config: Optional[Dict] = None, device="cuda:0", dtype=torch.bfloat16, seed: int = 0,
                 state_dict: Optional[Dict] = None, llm_max_length: int = 2048):
        from transformers import LlamaConfig, LlamaForCausalLM
        cfg = LlamaConfig(**(config or llama2_7b_config()))
        cfg._attn_implementation = "sdpa"
        torch.manual_seed(seed)
        with torch.device(device):
            self.llm = LlamaForCausalLM(cfg)
        self.llm = self.llm.to(dtype).eval().requires_grad_(False)
        if state_dict is not None:
            self.llm.load_state_dict(state_dict)
        self.device, self.dtype, self.llm_max_length = torch.device(device), dtype, llm_max_length
        self.embed_dim = cfg.hidden_size
        self.config = cfg

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
