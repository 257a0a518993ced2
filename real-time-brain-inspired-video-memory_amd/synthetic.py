"""Seeded synthetic inputs for the hot path: encoder weights, frames, memory rows, queries.

There is no network for checkpoints or datasets, and the reference ships neither encoder weights nor stored
embeddings (data/exports/*.json carry no ``embedding`` property), so benches and parity tests run on values drawn
from a counter-based generator (numpy Philox) keyed by (seed, tensor name): any tensor can be regenerated on any
box, in any order, from the seed alone.
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np


def _gen(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[int(seed), zlib.crc32(name.encode())]))


def normal(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    return (_gen(seed, name).standard_normal(shape, dtype=np.float32) * np.float32(std) + np.float32(mean))


def encoder_weight_shapes(spec: Dict) -> Dict[str, tuple]:
    H, P, S, M = spec["hidden"], spec["patch"], spec["image"], spec["mlp"]
    n_tok = (S // P) ** 2 + 1
    shapes = {
        "patch_w": (H, 3 * P * P), "patch_b": (H,), "cls": (H,), "pos": (n_tok, H),
        "ln_g": (H,), "ln_b": (H,),
    }
    if spec["pre_ln"]:
        shapes["pre_ln_g"] = (H,)
        shapes["pre_ln_b"] = (H,)
    if spec.get("proj_dim", 0):
        shapes["proj_w"] = (spec["proj_dim"], H)
    for l in range(spec["layers"]):
        shapes.update({
            f"l{l}.ln1_g": (H,), f"l{l}.ln1_b": (H,),
            f"l{l}.qkv_w": (3 * H, H), f"l{l}.qkv_b": (3 * H,),
            f"l{l}.proj_w": (H, H), f"l{l}.proj_b": (H,),
            f"l{l}.ln2_g": (H,), f"l{l}.ln2_b": (H,),
            f"l{l}.fc1_w": (M, H), f"l{l}.fc1_b": (M,),
            f"l{l}.fc2_w": (H, M), f"l{l}.fc2_b": (H,),
        })
    return shapes


def encoder_weights(spec: Dict, seed: int = 42, std: float = 0.02, tail: str = "normal") -> Dict[str, np.ndarray]:
    """fp32 master weights: linear/conv/cls/pos/bias ~ N(0, std^2); LayerNorm gamma = 1 + N(0, std^2).
    tail="heavy": the same variances from a Student-t with 4 degrees of freedom (excess kurtosis: single weights at
    6-10 sigma), plus six "outlier" hidden channels whose LayerNorm gains are 6x - the heavy-tailed weight AND
    activation statistics trained vision transformers show, which a 16-bit storage format is most sensitive to."""
    out = {}
    heavy = tail == "heavy"
    for name, shape in encoder_weight_shapes(spec).items():
        is_gamma = name.endswith("_g")
        if heavy and not is_gamma:
            t = _gen(seed, name).standard_t(4.0, size=shape).astype(np.float32) * np.float32(std / np.sqrt(2.0))
            out[name] = t
        else:
            out[name] = normal(seed, name, shape, std=std, mean=1.0 if is_gamma else 0.0)
        if heavy and is_gamma:
            ch = _gen(seed, "outlier_channels").choice(shape[0], size=6, replace=False)
            out[name][ch] *= np.float32(6.0)
    if not spec["patch_bias"]:
        out["patch_b"] = np.zeros_like(out["patch_b"])
    return out


def frames_u8(seed: int, n: int, height: int, width: int) -> np.ndarray:
    """uint8 [n, H, W, 3] BGR frames, i.i.d. uniform 0..255 (what cv2.VideoCapture.read hands over,
    reference src/pipeline/vlm_extractor.py:110-117)."""
    return _gen(seed, f"frames{n}x{height}x{width}").integers(0, 256, size=(n, height, width, 3), dtype=np.uint8)


def unit_rows(seed: int, name: str, rows: int, dim: int) -> np.ndarray:
    """fp32 [rows, dim] L2-normalised N(0,1) vectors (memory rows / queries)."""
    x = _gen(seed, name).standard_normal((rows, dim), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x
