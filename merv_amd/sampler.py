"""
Frame sampler of the MERV front door, re-pointed at libmerv_hip.so's host-only entry points:

  frame_indices      <- np.linspace(..., dtype=int) in load_video (merv/preprocessing/datasets/datasets.py:131-141)
  temporal_subsample <- video[:: max(num_frames) // nf]           (merv/models/vidlms/merv.py:803-806)

Frame *decoding* stays on the CPU with whatever decoder the caller has (decord in the reference); only the index
math -- the part that must be bit-exact -- lives here.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional

from . import _lib


def frame_indices(video_num_frames: int, avg_fps: float, clip_start_sec: Optional[float] = 0.0,
                  clip_end_sec: Optional[float] = None, num_frames: int = 8, end_frame: Optional[int] = None) -> List[int]:
    """Indices load_video() would hand to decord's get_batch(). `None` arguments mean what they mean in the reference."""
    lib = _lib.load()
    out = (C.c_int64 * max(num_frames, 1))()
    start = 0.0 if clip_start_sec is None else float(clip_start_sec)
    end = float("nan") if clip_end_sec is None else float(clip_end_sec)
    ef = -1 if (end_frame is None or end_frame < 0) else int(end_frame)
    rc = lib.merv_frame_indices(int(video_num_frames), float(avg_fps), start, end, ef, int(num_frames), out)
    _lib.check(rc, "merv_frame_indices")
    return [int(out[i]) for i in range(num_frames)]


def temporal_subsample(loaded_frames: int, max_nf: int, nf: int) -> List[int]:
    lib = _lib.load()
    idx = (C.c_int32 * max(loaded_frames, 1))()
    n = C.c_int32()
    rc = lib.merv_temporal_subsample(int(loaded_frames), int(max_nf), int(nf), idx, C.byref(n))
    _lib.check(rc, "merv_temporal_subsample")
    return [int(idx[i]) for i in range(n.value)]
