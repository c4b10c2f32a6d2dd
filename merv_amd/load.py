"""
`load_vid()` / `available_models()` / `available_model_names()` / `get_model_description()` with the reference's
signatures and on-disk layout (merv/models/load_vid.py:27-127):

    <run_dir>/config.json                       {"model": {model_id, arch_specifier, feature_fusion, video_backbone_ids,
                                                 llm_backbone_id, image_resize_strategy, llm_max_length, num_frames,
                                                 projector_token_length, visual_feature_length, ...}}
    <run_dir>/checkpoints/latest-checkpoint.pt  {"model": {"projectors": {...}, "feature_fusion": {...},
                                                 "llm_backbone": {...}}}        (fsdp.py:136-141, merv.py:272-289)

What differs, because this machine has no network: the reference pulls the run from the HF Hub when given a registry
id and each frozen encoder from timm / HF inside its backbone constructor. Here a registry id resolves to
`<cache_dir or $MERV_HOME>/<model_id>/`, and encoder parameters come from local state-dict files in the upstream key
layouts (timm, HF ViViT / DINOv2 / SigLIP, LanguageBind -- merv_amd/weights.py converts them):

    <encoder_dir>/<video_backbone_id>.{pt,pth,bin,safetensors}

with `<encoder_dir>` = the `encoder_weights` argument (a directory, or a dict id -> state dict / path / "random"),
else `$MERV_ENCODER_WEIGHTS`, else `<run_dir>/encoders/`. A missing encoder file is an error, never a silent random
initialisation.
"""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Dict, List, Mapping, Optional, Union

import torch

from .backbones import get_video_backbone_and_transform
from .llm import get_llm_backbone_and_tokenizer
from .registry import GLOBAL_REGISTRY, MODEL_REGISTRY, resolve_model_config
from .vidlm import MERV

_SUFFIXES = (".pt", ".pth", ".bin", ".safetensors")


def available_models() -> List[str]:
    return list(MODEL_REGISTRY.keys())


def available_model_names() -> List[str]:
    return list(GLOBAL_REGISTRY.items())


def get_model_description(model_id_or_name: str) -> str:
    if model_id_or_name not in GLOBAL_REGISTRY:
        raise ValueError(f"Couldn't find `{model_id_or_name = }; check `merv_amd.load.available_model_names()`")
    print(json.dumps(description := GLOBAL_REGISTRY[model_id_or_name]["description"], indent=2))
    return description


def _read_state_dict(path: Path) -> Dict[str, torch.Tensor]:
    if path.suffix == ".safetensors":
        from safetensors.torch import load_file
        return load_file(str(path))
    sd = torch.load(path, map_location="cpu", weights_only=True)
    for key in ("state_dict", "model"):  # common wrappers around a plain state dict
        if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
            sd = sd[key]
    return sd


def _encoder_weights(video_backbone_ids: List[str], source, run_dir: Path) -> List:
    if isinstance(source, Mapping):
        missing = [i for i in video_backbone_ids if i not in source]
        if missing:
            raise FileNotFoundError(f"encoder_weights has no entry for {missing}")
        return [_read_state_dict(Path(source[i])) if isinstance(source[i], (str, Path)) and source[i] != "random" else source[i]
                for i in video_backbone_ids]
    enc_dir = Path(source) if source is not None else Path(os.environ.get("MERV_ENCODER_WEIGHTS", run_dir / "encoders"))
    out = []
    for vid in video_backbone_ids:
        hits = [enc_dir / (vid + sfx) for sfx in _SUFFIXES if (enc_dir / (vid + sfx)).exists()]
        if not hits:
            raise FileNotFoundError(f"no weights for video backbone `{vid}` under `{enc_dir}` (expected {vid}.pt / .safetensors "
                                    f"in the upstream key layout; the hub download of the reference is not available here)")
        out.append(_read_state_dict(hits[0]))
    return out


def load_vid(model_id_or_path: Union[str, Path], hf_token: Optional[str] = None, cache_dir: Optional[Union[str, Path]] = None,
             get_model_cfg: bool = False, *, encoder_weights=None, llm_config: Optional[Dict] = None, tokenizer=None,
             tokenizer_path: Optional[Union[str, Path]] = None, device="cuda:0") -> MERV:
    """Loads a pretrained MERV from local disk (load_vid.py:46-127). Keyword-only arguments are this build's additions:
    `encoder_weights` (see module docstring), `llm_config` (geometry override for reduced-size runs; default = the
    registry geometry of `llm_backbone_id`), `tokenizer` / `tokenizer_path` (local tokenizer; without one `generate()`
    takes and returns token ids)."""
    if os.path.isdir(model_id_or_path):
        run_dir = Path(model_id_or_path)
    else:
        if model_id_or_path not in GLOBAL_REGISTRY:
            raise ValueError(f"Couldn't find `{model_id_or_path = }; check `merv_amd.load.available_model_names()`")
        model_id = GLOBAL_REGISTRY[model_id_or_path]["model_id"]
        root = cache_dir if cache_dir is not None else os.environ.get("MERV_HOME")
        if root is None or not (Path(root) / model_id).is_dir():
            raise FileNotFoundError(f"`{model_id}` is a registry id, but there is no hub access here: place the run under "
                                    f"`<cache_dir or $MERV_HOME>/{model_id}/` (config.json + checkpoints/latest-checkpoint.pt)")
        run_dir = Path(root) / model_id
    config_json = run_dir / "config.json"
    checkpoint_pt = run_dir / "checkpoints" / "latest-checkpoint.pt"
    assert config_json.exists(), f"Missing `config.json` for `{run_dir = }`"
    assert checkpoint_pt.exists(), "Missing checkpoint!"

    with open(config_json, "r") as f:
        model_cfg = resolve_model_config(json.load(f)["model"])

    video_backbones, _ = get_video_backbone_and_transform(
        model_cfg["video_backbone_ids"], image_resize_strategy=model_cfg["image_resize_strategy"],
        num_frames=model_cfg["num_frames"], weights=_encoder_weights(model_cfg["video_backbone_ids"], encoder_weights, run_dir),
        device=device)
    if tokenizer is None and tokenizer_path is None and (run_dir / "tokenizer").is_dir():
        tokenizer_path = run_dir / "tokenizer"
    llm_backbone, tokenizer = get_llm_backbone_and_tokenizer(
        model_cfg["llm_backbone_id"], llm_max_length=model_cfg["llm_max_length"], hf_token=hf_token, inference_mode=True,
        config=llm_config, tokenizer=tokenizer, tokenizer_path=None if tokenizer_path is None else str(tokenizer_path),
        device=device)
    vidlm = MERV.from_pretrained(checkpoint_pt, model_cfg["model_id"], video_backbones, llm_backbone, tokenizer=tokenizer,
                                 arch_specifier=model_cfg["arch_specifier"], feature_fusion=model_cfg["feature_fusion"],
                                 visual_feature_length=model_cfg["visual_feature_length"],
                                 projector_token_length=model_cfg["projector_token_length"])
    return (vidlm, model_cfg) if get_model_cfg else vidlm
