"""
MERVVisual: the visual branch of the reference's MERV VidLM (merv/models/vidlms/merv.py) with the same constructor
vocabulary (`arch_specifier`, `feature_fusion`, `projector_token_length`, `visual_feature_length`), the same module
names (`projectors`, `feature_fusion`) and therefore the same checkpoint keys (merv.py:272-289), running on the HIP
path. The LLM stays outside (PyTorch-ROCm): `forward_visual` returns what MERV.forward hands to
`llm_backbone(inputs_embeds=...)` (merv.py:723-734).
"""
from __future__ import annotations

import re
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .backbones import VideoBackbone
from .projector import AveragePooling3DProjector, CrossAttentionAdapterLearnableQuery, splice
from .visual_path import MervVisualPath

IGNORE_INDEX = -100


def bos_token_length(llm_backbone, tokenizer=None) -> int:
    """merv.py:520-521: `1 if self.llm_backbone.tokenizer.bos_token is not None else 0` ("QWEN2.5-7B-INSTRUCT has no BOS
    token"). Decided from the TOKENIZER, as the reference does -- Qwen2.5's config.json carries bos_token_id = 151643 although
    its tokenizer prepends nothing, so config.bos_token_id is not the test. Order: the tokenizer's `bos_token` attribute when a
    tokenizer is attached (HF tokenizers, HFTokenizerAdapter), else the backbone's per-family `prepends_bos` flag (LLM registry;
    False for the qwen2 family), else True."""
    tok = tokenizer if tokenizer is not None else getattr(llm_backbone, "tokenizer", None)
    inner = getattr(tok, "tok", tok)  # HFTokenizerAdapter wraps the HF object as `.tok`
    if inner is not None and hasattr(inner, "bos_token"):
        return 1 if inner.bos_token is not None else 0
    return 1 if getattr(llm_backbone, "prepends_bos", True) else 0


class MERVVisual(nn.Module):
    def __init__(self, video_backbones: Sequence[VideoBackbone], llm_dim: int = 4096, arch_specifier: str = "3davg+linear",
                 feature_fusion: Optional[str] = "cross_attention_avg_lq", projector_token_length: int = 64,
                 visual_feature_length: int = 1024, concurrent_streams: bool = True) -> None:
        super().__init__()
        self.video_backbones = list(video_backbones)  # frozen, not registered as sub-modules (merv.py:315-381)
        # The visual path drives the patch-token featurizers directly; a backbone id whose forward() selects something else
        # (class token, per-frame average, pooled head: materialize.py:31-73) goes through its own forward() and the projector
        # grid the reference derives from it (merv.py:576-585: [B, temporal_resolution, spatial_resolution, C], then
        # nn_utils.py:320-330 with H = int(sqrt(spatial_resolution))). Selections the reference's own reshape / rearrange rejects
        # (a class token in front of the patches: F * 256 + 1 tokens; 257 tokens per frame: not H x W) are rejected here as well.
        self._selectors = []
        for vb in self.video_backbones:
            if getattr(vb, "selects_spec_patches", True):
                self._selectors.append(None)
                continue
            S, T = vb.spatial_resolution, vb.temporal_resolution
            side = int(S ** 0.5)
            if side * side != S:
                raise ValueError(f"`{vb.identifier}`: {S} tokens per frame do not form an H x W grid -- the reference's "
                                 "AveragePooling3DProjector fails on it too (einops 'B F (H W) C', nn_utils.py:322-326)")
            if "classemb-at-first" in vb.identifier:  # forward() returns T * S + 1 tokens (the class token in front of all patches)
                raise ValueError(f"`{vb.identifier}`: forward() returns {T * S + 1} tokens, which do not reshape to [B, {T}, {S}, C] -- the "
                                 "reference's MERV.forward fails on it too (merv.py:576-585)")
            self._selectors.append((vb.forward, T, side))
        self.feature_fusion_type = feature_fusion
        torch.manual_seed(self.video_backbones[0].embed_dim)  # merv.py:87: projector-init consistency
        self.arch_specifier = arch_specifier
        if not arch_specifier.endswith("linear"):
            raise ValueError(f"MERV with `{arch_specifier = }` is not supported on the HIP path (only '...+linear')!")
        parts = arch_specifier.split("+")
        if "3davg" not in parts:
            raise ValueError(f"MERV with `{arch_specifier = }` is not supported on the HIP path (only '3davg')!")
        factor = 1
        if "frame" in arch_specifier:  # merv.py:114-117
            factor = int(re.search(r"frame(\d+)", arch_specifier).group(1))
        if factor != 1:
            raise NotImplementedError("temporal down-sampling projectors (frameN) are not used by merv-full")
        out_size = int(projector_token_length**0.5)
        assert projector_token_length == out_size**2, "projector_token_length should be square number"
        self.tokens_resampled = True
        self.projectors = nn.ModuleList([
            AveragePooling3DProjector(vb.embed_dim, llm_dim, output_frames=vb.temporal_resolution // factor,
                                      output_size=out_size, mlp_type="linear") for vb in self.video_backbones])
        if len(self.video_backbones) > 1:  # merv.py:175-185
            assert all(p.output_token_length * p.output_frame_length in [1, visual_feature_length] for p in self.projectors), (
                "Output token length is not consistent across all projectors!" f" visual_feature_length={visual_feature_length}.")
        else:
            visual_feature_length = self.projectors[0].output_token_length * self.projectors[0].output_frame_length
        self.visual_feature_length = visual_feature_length
        if feature_fusion == "cross_attention_avg_lq":  # merv.py:213-216
            self.feature_fusion = CrossAttentionAdapterLearnableQuery(embed_dim=3072, llm_dim=llm_dim,
                                                                      token_length=visual_feature_length, averagetoken=True)
        elif feature_fusion is None:
            self.feature_fusion = None
        else:
            raise NotImplementedError(f'feature_fusion "{feature_fusion}" is not wired on the HIP path')
        self.concurrent = concurrent_streams and len(self.video_backbones) > 1
        self._path: Optional[MervVisualPath] = None
        self._path_versions = None

    # -- checkpoint layout of the reference: {"model": {"projectors": {...}, "feature_fusion": {...}, "llm_backbone": ...}}
    def load_from_checkpoint_dict(self, model_state_dict: Dict) -> None:
        sd = dict(model_state_dict)
        if "projector" in sd:  # legacy single-projector checkpoints (merv.py:273-274)
            sd["projectors"] = {"0." + k: v for k, v in sd["projector"].items()}
        self.projectors.load_state_dict(sd["projectors"])
        if self.feature_fusion is not None:
            if "feature_fusion" in sd:
                self.feature_fusion.load_state_dict(sd["feature_fusion"])
            elif "adapter" in sd:
                self.feature_fusion.load_state_dict(sd["adapter"])
            self.feature_fusion._u = None
        for p in self.projectors:
            p._dev = None
        self._path = None  # the path holds bf16 device copies of the projector weights and the folded fusion vector

    # -- the ONE orchestration of the visual path (merv_amd/visual_path.py): built lazily from this module's backbones,
    #    projectors and fusion module; rebuilt when their parameters change (checkpoint load, .to())
    def _param_versions(self):
        ps = [q for p in self.projectors for q in p.parameters()]
        if self.feature_fusion is not None:
            ps += list(self.feature_fusion.parameters())
        return tuple((q.data_ptr(), q._version) for q in ps)

    def visual_path(self, device=None) -> MervVisualPath:
        dev = torch.device(device) if device is not None else self.video_backbones[0].featurizer.device
        ver = self._param_versions()  # in-place updates (optimizer steps, load_state_dict) bump Tensor._version
        if self._path is None or self._path.device != dev or ver != self._path_versions:
            proj_w = [(p.projector.projector.weight, p.projector.projector.bias) for p in self.projectors]
            if self.feature_fusion is not None:
                self.feature_fusion._u = None
            if self._path is not None and self._path.device == dev:
                self._path.set_parameters(proj_w, self.feature_fusion)  # keep workspaces and buffers
            else:
                self._path = MervVisualPath([vb.spec for vb in self.video_backbones], None, proj_w, self.feature_fusion, dev,
                                            out_size=self.projectors[0].output_size, concurrent_streams=self.concurrent,
                                            encoders=[vb.featurizer for vb in self.video_backbones], selectors=self._selectors)
            self._path_versions = ver
        return self._path

    def _apply(self, fn, *args, **kw):  # .to() / .bfloat16() / .cuda(): device copies held by the path go stale
        self._path = None
        return super()._apply(fn, *args, **kw)

    @torch.no_grad()
    def encode(self, video_values: Sequence[torch.Tensor]) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """merv.py:562-612: encoders -> [B,T,S,C] -> projectors -> fusion. Returns (fused [B, L, llm], weights [B,E]|None).
        Runs MervVisualPath.forward -- the same code bench.py times: persistent per-(encoder, batch) buffers, four encoder
        streams, no per-call allocation. The returned tensors are the path's buffers (overwritten by the next call)."""
        if len(video_values) != len(self.video_backbones):
            raise RuntimeError("Invalid `forward()` call!")  # merv.py:540-541
        if self.feature_fusion_type is None and len(self.video_backbones) != 1:
            raise TypeError("argument of type 'NoneType' is not iterable")  # reference behaviour, merv.py:607 (App. B.4)
        return self.visual_path(video_values[0].device).forward(video_values)

    @torch.no_grad()
    def forward_visual(self, video_values: Sequence[torch.Tensor], input_embeddings: torch.Tensor,
                       attention_mask: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                       bos_token_length: int = 1):
        """merv.py:595-664: fused tokens spliced after BOS; attention mask True / labels IGNORE_INDEX over the visual span."""
        fused, w = self.encode(video_values)
        emb = splice(input_embeddings, fused, bos_token_length)
        b = bos_token_length
        am = lab = None
        if attention_mask is not None:
            vis = torch.full(fused.shape[:2], True, dtype=attention_mask.dtype, device=attention_mask.device)
            am = torch.cat([attention_mask[:, :b], vis, attention_mask[:, b:]], dim=1)
        if labels is not None:
            vis = torch.full(fused.shape[:2], IGNORE_INDEX, dtype=labels.dtype, device=labels.device)
            lab = torch.cat([labels[:, :b], vis, labels[:, b:]], dim=1)
        return emb, am, lab, w


class MERV(MERVVisual):
    """MERVVisual + an LLM backbone: the reference's front door `generate(video, prompt_text, num_frames, **kwargs)`
    (merv.py:778-830) with an explicit prefill + decode loop.

    `tokenizer` is any callable object with `__call__(text) -> list[int]` (incl. BOS) and `decode(ids) -> str`; HF
    tokenizers are adapted by `HFTokenizerAdapter`. (The Llama tokenizer files cannot be fetched here.)"""

    model_family = "merv"

    def __init__(self, video_backbones, llm_backbone, tokenizer=None, model_id: str = "merv", **kw) -> None:
        super().__init__(video_backbones, llm_dim=llm_backbone.embed_dim, **kw)
        self.model_id = model_id
        self.llm_backbone = llm_backbone
        self.tokenizer = tokenizer

    @classmethod
    def from_pretrained(cls, pretrained_checkpoint, model_id: str, video_backbones, llm_backbone,
                        enable_mixed_precision_training: bool = True, arch_specifier: str = "3davg+linear",
                        feature_fusion: Optional[str] = None, visual_feature_length: Optional[int] = -1,
                        projector_token_length: Optional[int] = -1, tokenizer=None) -> "MERV":
        """merv.py:246-299: build, then load `projectors`, `llm_backbone` and `feature_fusion` (or legacy `adapter`) from
        the checkpoint's "model" dict; everything frozen, eval mode."""
        vidlm = cls(video_backbones, llm_backbone, tokenizer=tokenizer, model_id=model_id, arch_specifier=arch_specifier,
                    feature_fusion=feature_fusion, visual_feature_length=visual_feature_length,
                    projector_token_length=projector_token_length)
        model_state_dict = torch.load(pretrained_checkpoint, map_location="cpu", weights_only=True)["model"]
        if "projector" in model_state_dict:
            model_state_dict["projectors"] = {"0." + k: v for k, v in model_state_dict["projector"].items()}
        assert "projectors" in model_state_dict and "llm_backbone" in model_state_dict, (
            "MERV `from_pretrained` expects checkpoint with keys for `projector` AND `llm_backbone`!"
            + f'{("projectors" in model_state_dict, "llm_backbone" in model_state_dict)}')
        if vidlm.feature_fusion is None:
            assert "feature_fusion" not in model_state_dict or len(model_state_dict["feature_fusion"]) == 0, \
                model_state_dict["feature_fusion"]
        vidlm.load_from_checkpoint_dict(model_state_dict)
        vidlm.llm_backbone.load_state_dict(model_state_dict["llm_backbone"])
        vidlm.to(llm_backbone.device)
        vidlm.requires_grad_(False)
        vidlm.eval()
        return vidlm

    def get_prompt_builder(self, system_prompt=None, chat: Optional[bool] = None):
        """merv.py:832-835: the LLM backbone picks the builder; `chat` is kept as an explicit override."""
        from .prompting import LLaMa2ChatPromptBuilder, PurePromptBuilder
        if chat is None and hasattr(self.llm_backbone, "prompt_builder_fn"):
            return self.llm_backbone.prompt_builder_fn(self.model_family, system_prompt=system_prompt)
        return (LLaMa2ChatPromptBuilder if chat else PurePromptBuilder)(self.model_family, system_prompt=system_prompt)

    @torch.inference_mode()
    def generate(self, video, prompt_text, num_frames, **kwargs):
        assert len(num_frames) == len(self.video_backbones), "Number of frames should match number of video backbones!"
        from .sampler import temporal_subsample
        from .video_io import load_video
        dev = self.llm_backbone.device
        if isinstance(prompt_text, str):
            if self.tokenizer is None:
                raise ValueError("generate(prompt_text: str) needs a tokenizer; pass token ids otherwise")
            input_ids = torch.tensor([self.tokenizer(prompt_text)], dtype=torch.long, device=dev)
        else:
            input_ids = torch.as_tensor(prompt_text, dtype=torch.long, device=dev).reshape(1, -1)
        clip_start_sec = kwargs.pop("clip_start_sec", 0.0)
        clip_end_sec = kwargs.pop("clip_end_sec", None)
        end_frame = kwargs.pop("end_frame", None)
        max_new = kwargs.pop("max_new_tokens", 32)
        gen = dict(max_new_tokens=max_new, do_sample=kwargs.pop("do_sample", False), temperature=kwargs.pop("temperature", 1.0),
                   top_k=kwargs.pop("top_k", 0) or 0, top_p=kwargs.pop("top_p", 1.0) or 1.0,
                   repetition_penalty=kwargs.pop("repetition_penalty", 1.0) or 1.0)
        # HF generate kwargs the reference forwards (merv.py:818-825) that change nothing here, or that this explicit
        # prefill + decode loop does not implement: the former are accepted, the latter fail loudly instead of silently
        # decoding differently from the reference's eval scripts
        for k in ("use_cache", "pad_token_id", "attention_mask", "is_image", "return_dict_in_generate"):
            kwargs.pop(k, None)
        if kwargs.pop("num_beams", 1) not in (None, 1):
            raise NotImplementedError("beam search (num_beams > 1) is not implemented by merv_amd's decode loop")
        # HF MinLengthLogitsProcessor counts the prompt's input_ids (the reference passes input_ids, merv.py:819): EOS is
        # barred while prompt + generated < min_length (every reference script passes min_length=1: a no-op)
        gen["min_new_tokens"] = max(int(kwargs.pop("min_new_tokens", 0) or 0), int(kwargs.pop("min_length", 0) or 0) - input_ids.shape[1])
        if kwargs:
            raise TypeError(f"generate(): unsupported generation arguments {sorted(kwargs)}")
        max_len = getattr(self.llm_backbone, "llm_max_length", None)
        if max_len:  # tokenizer(prompt_text, truncation=True): HF truncates to the tokenizer's model_max_length (merv.py:786)
            input_ids = input_ids[:, :max_len]
        if video is not None and isinstance(video, (str, Path)) and ".jpg" in str(video):  # still image (merv.py:787-793)
            import numpy as np
            from PIL import Image
            image = Image.open(str(video)).convert("RGB")
            frames = torch.from_numpy(np.array(image).transpose(2, 0, 1)[None,].repeat(max(num_frames), 0)).to(dev)
            video_values = [vb.video_transform(frames[:: max(num_frames) // nf].contiguous()).unsqueeze(0)
                            for vb, nf in zip(self.video_backbones, num_frames)]
        elif video is not None:
            frames = load_video(video, clip_start_sec=clip_start_sec, clip_end_sec=clip_end_sec, num_frames=max(num_frames),
                                end_frame=end_frame).to(dev)  # uint8 [T,3,H,W]
            video_values = []
            for vb, nf in zip(self.video_backbones, num_frames):
                idx = temporal_subsample(frames.shape[0], max(num_frames), nf)  # video[:: max(num_frames) // nf]
                video_values.append(vb.video_transform(frames[idx].contiguous()).unsqueeze(0))
        else:  # merv.py:807-811
            video_values = [torch.zeros(vb.default_video_resolution, device=dev).unsqueeze(0) for vb in self.video_backbones]
        emb = self.llm_backbone.embed_input_ids(input_ids)
        bos = bos_token_length(self.llm_backbone, self.tokenizer)  # merv.py:520-521
        fused_emb, _, _, weights = self.forward_visual(video_values, emb, bos_token_length=bos)
        ids = self.llm_backbone.generate_from_embeds(fused_emb, eos_token_id=getattr(self.llm_backbone.config, "eos_token_id", None),
                                                     prompt_ids=input_ids, **gen)  # HF penalises the prompt's ids too (merv.py:819)
        self.last_fusion_weights = weights
        if self.tokenizer is not None and hasattr(self.tokenizer, "decode"):
            return self.tokenizer.decode(ids[0].tolist()).strip()
        return ids
