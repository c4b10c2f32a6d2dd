"""
load_video with the reference's signature (merv/preprocessing/datasets/datasets.py:35-160). The index math is the
bit-exact library call (merv_amd/sampler.py); decoding stays on the CPU with what this image has:

  * a pre-decoded clip: `(frames uint8 [N,H,W,3] or [N,3,H,W], avg_fps)` -- what decord's VideoReader would expose;
  * an animated .gif via PIL (the reference's own gif branch, datasets.py:117-123);
  * .mp4 / .avi need decord (absent here): a clear ImportError, never a silent fallback.
Returns uint8 [T, 3, H, W] like the reference.
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Optional, Tuple, Union

import numpy as np
import torch

from .sampler import frame_indices


def _to_tchw(frames: torch.Tensor) -> torch.Tensor:
    if frames.dim() != 4:
        raise ValueError("frames must be [N,H,W,3] or [N,3,H,W]")
    return frames.permute(0, 3, 1, 2) if frames.shape[-1] == 3 and frames.shape[1] != 3 else frames


def load_video(video_path: Union[str, Path, Tuple[torch.Tensor, float]], decode_backend: str = "decord",
               clip_start_sec: Optional[float] = 0.0, clip_end_sec: Optional[float] = None, num_frames: int = 8,
               end_frame: Optional[int] = None) -> torch.Tensor:
    if clip_start_sec is not None and math.isnan(clip_start_sec):  # datasets.py:46-52
        clip_start_sec = 0
    if clip_end_sec is not None and math.isnan(clip_end_sec):
        clip_end_sec = None
    if decode_backend != "decord":
        raise NameError(f"Unknown decode backend: {decode_backend}")
    if isinstance(video_path, tuple):  # pre-decoded clip
        frames, avg_fps = video_path
        frames = _to_tchw(torch.as_tensor(frames))
        ids = frame_indices(frames.shape[0], float(avg_fps), clip_start_sec, clip_end_sec, num_frames, end_frame)
        return frames[torch.as_tensor(ids, dtype=torch.long)].contiguous()
    path = Path(video_path)
    if path.is_dir():  # directories of decoded frames (datasets.py:59-112): VLEP at 3 fps (*.jpg), ShareGPT (*.jpeg)
        from PIL import Image
        low = str(path).lower()
        if "vlep" in low:
            images = sorted(str(p) for p in path.glob("*.jpg"))
            assert len(images) > 0, f"video directory contains no frames to load video - {path}"
            ids = frame_indices(len(images), 3.0, clip_start_sec, clip_end_sec, num_frames, None)  # fps_in_dir = 3 (:63)
        elif "sharegpt" in low:
            images = sorted(str(p) for p in path.glob("*.jpeg"))
            assert len(images) > 0, f"video directory contains no frames to load video - {path}"
            ids = frame_indices(len(images), 1.0, 0.0, None, num_frames, len(images) - 1)  # np.linspace(0, N-1, n, dtype=int) (:96)
        else:
            raise NotImplementedError  # as the reference (:112)
        # the reference decodes with cv2.imread + BGR->RGB; PIL decodes the same JPEG to the same RGB values
        frames = np.stack([np.array(Image.open(images[int(i)]).convert("RGB")) for i in ids], 0)
        return torch.from_numpy(frames).permute(0, 3, 1, 2).contiguous()
    if path.suffix == ".gif":
        from PIL import Image, ImageSequence
        im = Image.open(str(path))
        frames = torch.from_numpy(np.stack([np.array(f.convert("RGB")) for f in ImageSequence.Iterator(im)], 0))
        n = frames.shape[0]
        ids = frame_indices(n, 1.0, 0.0, None, num_frames, n - 1)  # np.linspace(0, N-1, num_frames, dtype=int)  (:121)
        return frames[torch.as_tensor(ids, dtype=torch.long)].permute(0, 3, 1, 2).contiguous()
    try:
        from decord import VideoReader, cpu  # noqa: F401
    except ImportError as e:
        raise ImportError("decoding video files needs `decord`, which this environment does not have; pass a "
                          "pre-decoded (frames, fps) pair or a .gif") from e
    vr = VideoReader(str(path), ctx=cpu(0))
    ids = frame_indices(len(vr), vr.get_avg_fps(), clip_start_sec, clip_end_sec, num_frames, end_frame)
    data = vr.get_batch(ids)
    data = torch.as_tensor(data.asnumpy() if hasattr(data, "asnumpy") else data)
    return data.permute(0, 3, 1, 2).contiguous()
