"""
Pretrained-model registry with the reference's ids and display names (merv/models/registry.py:9-94) and the per-id
model configuration the reference keeps in draccus dataclasses (merv/conf/models.py:101-187): what `load_vid()` needs
to rebuild a VidLM from `config.json["model"]`, and the defaults a config written by an older run may omit.
"""
from __future__ import annotations

from typing import Dict, List


def _entry(model_id: str, name: str, procedure: str, visual: str) -> Dict:
    return {"model_id": model_id, "names": [name],
            "description": {"name": name, "optimization_procedure": procedure, "visual_representation": visual,
                            "image_processing": "Letterbox", "language_model": "Llama 2 7B", "datasets": ["Video-LLaVA"],
                            "train_epochs": 1}}


MODEL_REGISTRY: Dict[str, Dict] = {
    "merv-frozen": _entry("merv-frozen", "MERV Frozen", "single-stage", "LanguageBind, DINO, SigLIP, ViViT"),
    "merv-full": _entry("merv-full", "MERV Full", "multi-stage", "LanguageBind, DINO, SigLIP, ViViT"),
    "languagebind-single": _entry("languagebind-single", "LanguageBind Single Encoder", "single-stage", "LanguageBind"),
    "dinov2-single": _entry("dinov2-single", "DINOv2 Single Encoder", "single-stage", "DINO"),
    "vivit-single": _entry("vivit-single", "ViViT Single Encoder", "single-stage", "ViViT"),
    "siglip-single": _entry("siglip-single", "SigLIP Single Encoder", "single-stage", "SigLIP"),
}

# (model id, display name) -> metadata, registry.py:92
GLOBAL_REGISTRY: Dict[str, Dict] = {name: v for k, v in MODEL_REGISTRY.items() for name in [k] + v["names"]}

_FOUR = ["languagebind-video-noclass", "dinov2-video-all-tokens", "vivit-google-b-all-no-cls-16frames",
         "siglip-vit-b16-224px-all-no-cls"]

# Inference-relevant fields of merv/conf/models.py:101-187 (optimisation hyper-parameters are not part of this path).
MODEL_CONFIGS: Dict[str, Dict] = {
    "merv-base": dict(arch_specifier="no-align+3davg+linear", feature_fusion="cross_attention_avg_lq",
                      video_backbone_ids=_FOUR, llm_backbone_id="llama2-7b-pure", image_resize_strategy="resize-naive",
                      llm_max_length=2048, num_frames=[16, 16, 32, 16], projector_token_length=64,
                      visual_feature_length=1024),
}
MODEL_CONFIGS["merv-frozen"] = dict(MODEL_CONFIGS["merv-base"])
MODEL_CONFIGS["merv-full"] = dict(MODEL_CONFIGS["merv-base"], arch_specifier="3davg+linear")
for _mid, _bb, _nf in [("languagebind-single", _FOUR[0], 16), ("dinov2-single", _FOUR[1], 16),
                       ("vivit-single", _FOUR[2], 32), ("siglip-single", _FOUR[3], 16)]:
    MODEL_CONFIGS[_mid] = dict(MODEL_CONFIGS["merv-base"], video_backbone_ids=[_bb], num_frames=[_nf])

INFERENCE_KEYS: List[str] = ["model_id", "arch_specifier", "feature_fusion", "video_backbone_ids", "llm_backbone_id",
                             "image_resize_strategy", "llm_max_length", "num_frames", "projector_token_length",
                             "visual_feature_length"]


class ModelCfg(dict):
    """The resolved model config: a dict that also reads as attributes (`model_cfg.num_frames`), the way the reference's
    draccus dataclass is used by its scripts (scripts/eval_mcq.py:152)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e


def resolve_model_config(cfg: Dict) -> Dict:
    """`ModelConfig.get_choice_class("merv-base")(**cfg)` + `__post_init__` (load_vid.py:73-79, models.py:88-97):
    merv-base defaults under the given fields, and an int `num_frames` inflated to one entry per backbone."""
    out = dict(MODEL_CONFIGS["merv-base"], model_id="merv-base")
    out.update({k: v for k, v in cfg.items() if k not in ("vidlm_id", "type")})
    if isinstance(out["num_frames"], int):
        out["num_frames"] = [out["num_frames"]] * len(out["video_backbone_ids"])
    return ModelCfg(out)
