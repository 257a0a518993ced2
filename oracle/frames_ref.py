"""CPU ORACLE (test infrastructure, NOT product code) for the frame side of the hot path.

* ``chunk_plan_ref``  restates the chunking integers of the reference extractor:
      src/pipeline/vlm_extractor.py:38-39 (chunk_size_frames, total_chunks), :44-46 (start/end),
      :101-107 (n = min(frames_per_chunk, end-start); np.linspace(start, end-1, n, dtype=int)),
      :57-59 (the "MM:SS-MM:SS" label).
* ``downscale_720_size_ref``  src/pipeline/vlm_extractor.py:114-116 (only the output SIZE; the pixels come from
      cv2.resize, which is not installed here -> pixel parity of that step is UNPINNED).
* ``preprocess_ref``  is BUILD-DEFINED (the reference has no 224/336 resize, mean/std or CHW step: its frames go
      to a remote service as JPEG, :121-128): bilinear, half-pixel centres, no antialias, edge clamp
      (== torch.nn.functional.interpolate(mode="bilinear", align_corners=False), cross-checked in
      tests/test_frames_oracle.py), BGR->RGB, x*(1/(255*std)) - mean/std in fp32, round to fp16/bf16.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def chunk_plan_ref(fps: float, total_frames: int, chunk_size_seconds: float, frames_per_chunk: int):
    chunk_size_frames = int(chunk_size_seconds * fps)
    total_chunks = max(1, int(total_frames / chunk_size_frames))
    plan = []
    for chunk_idx in range(total_chunks):
        start_frame = chunk_idx * chunk_size_frames
        end_frame = min((chunk_idx + 1) * chunk_size_frames, total_frames)
        n = min(frames_per_chunk, end_frame - start_frame)
        if n <= 0:
            continue
        idx = np.linspace(start_frame, end_frame - 1, n, dtype=int)
        start_time = start_frame / fps
        end_time = end_frame / fps
        time_str = (f"{int(start_time // 60):02d}:{int(start_time % 60):02d}-"
                    f"{int(end_time // 60):02d}:{int(end_time % 60):02d}")
        plan.append(dict(chunk_idx=chunk_idx, start=start_frame, end=end_frame,
                         indices=[int(i) for i in idx], time=time_str))
    return total_chunks, plan


def downscale_720_size_ref(height: int, width: int) -> Tuple[int, int]:
    if height > 720:
        scale = 720 / height
        return 720, int(width * scale)
    return height, width


def _axis_taps(n_in: int, n_out: int):
    scale = np.float32(n_in) / np.float32(n_out)
    dst = np.arange(n_out, dtype=np.float32)
    src = scale * (dst + np.float32(0.5)) - np.float32(0.5)
    src = np.maximum(src, np.float32(0.0))
    i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    lam = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, lam


def preprocess_ref(frames_bgr_u8: np.ndarray, out_size: int, mean, std, layout: str = "chw", patch: int = 0,
                   k_pad: int = 0) -> np.ndarray:
    """uint8 [B,H,W,3] BGR -> fp32 [B,3,S,S] ('chw') or [B,(S/p)^2,k_pad] ('patches', (c,py,px) order, zero pad)."""
    f = np.asarray(frames_bgr_u8)
    B, H, W, _ = f.shape
    S = out_size
    y0, y1, ly = _axis_taps(H, S)
    x0, x1, lx = _axis_taps(W, S)
    rgb = f[..., ::-1].astype(np.float32)  # BGR -> RGB
    one = np.float32(1.0)
    p00 = rgb[:, y0][:, :, x0]
    p01 = rgb[:, y0][:, :, x1]
    p10 = rgb[:, y1][:, :, x0]
    p11 = rgb[:, y1][:, :, x1]
    lxb = lx[None, None, :, None]
    lyb = ly[None, :, None, None]
    top = (one - lxb) * p00 + lxb * p01
    bot = (one - lxb) * p10 + lxb * p11
    v = (one - lyb) * top + lyb * bot  # [B,S,S,3]
    a = (np.float32(1.0) / (np.float32(255.0) * np.asarray(std, np.float32))).astype(np.float32)
    b = (-np.asarray(mean, np.float32) / np.asarray(std, np.float32)).astype(np.float32)
    v = v * a + b
    chw = np.ascontiguousarray(v.transpose(0, 3, 1, 2))
    if layout == "chw":
        return chw
    g = S // patch
    x = chw.reshape(B, 3, g, patch, g, patch).transpose(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * patch * patch)
    k_pad = k_pad or x.shape[-1]
    out = np.zeros((B, g * g, k_pad), np.float32)
    out[..., : x.shape[-1]] = x
    return out
