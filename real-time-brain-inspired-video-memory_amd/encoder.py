"""Frame encoder + embedder adapters over libvidmem (csrc/encoder.hip, gemm.hip, attention.hip, context.hip).

Reference call sites this stands in for:
  * ``VLMExtractor._call_vlm_api``  src/pipeline/vlm_extractor.py:130-185  (frames -> remote vision model)
  * ``OpenAIEmbeddings(...).aembed_query``  src/components/neo4j_handler.py:27-31,333;
    src/components/pre_llm_injector.py:207-221; src/pipeline/retriever_hybrid.py:290,474,494
    (one vector per call; any object with ``async aembed_query(x) -> List[float]`` is an embedder there).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .memory import _torch_dtype


def _weight_order(spec: Dict) -> List[str]:
    names = ["patch_w", "patch_b", "cls", "pos", "pre_ln_g", "pre_ln_b", "ln_g", "ln_b", "proj_w"]
    for l in range(spec["layers"]):
        names += [f"l{l}.{n}" for n in ("ln1_g", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ln2_g", "ln2_b",
                                        "fc1_w", "fc1_b", "fc2_w", "fc2_b")]
    return names


_MATRICES = ("patch_w", "proj_w", "qkv_w", "fc1_w", "fc2_w")


class FrameEncoder:
    """uint8 BGR frames -> L2-normalised 16-bit embeddings, everything on the device."""

    def __init__(self, spec: Dict, weights: Dict[str, np.ndarray], dtype: str = "f16", device: int = 0,
                 schedule: str = "auto", micro_batch: int = 0, last_layer: int = 3):
        """schedule / micro_batch / last_layer: include/vidmem.h vm_encoder_set_option (same embeddings for every
        value; "auto" runs calls of two or more passes on two internal streams)."""
        self.spec = dict(spec)
        self.ctx = _lib.Context.get(device)
        self.L = self.ctx.L
        self.device = torch.device("cuda", device)
        self.dtype_name = dtype
        self.dtype = _torch_dtype(dtype)
        self.patch_k = (3 * spec["patch"] ** 2 + 63) // 64 * 64
        desc = _lib.EncoderDesc(
            image=spec["image"], patch=spec["patch"], hidden=spec["hidden"], layers=spec["layers"],
            heads=spec["heads"], mlp=spec["mlp"],
            act=_lib.VM_ACT_QUICK_GELU if spec["act"] == "quick_gelu" else _lib.VM_ACT_GELU,
            pre_ln=int(bool(spec["pre_ln"])), patch_bias=int(bool(spec["patch_bias"])),
            proj_dim=int(spec.get("proj_dim", 0)), dtype=_lib.DTYPES[dtype], ln_eps=float(spec["ln_eps"]))
        staged = []
        ptrs = []
        for name in _weight_order(spec):
            arr = weights.get(name)
            if arr is None:
                ptrs.append(None)
                continue
            t = torch.as_tensor(np.asarray(arr), dtype=torch.float32)
            if name.split(".")[-1] in _MATRICES:
                if name == "patch_w" and t.shape[1] < self.patch_k:
                    t = torch.nn.functional.pad(t, (0, self.patch_k - t.shape[1]))
                t = t.to(self.dtype)
            t = t.to(self.device).contiguous()
            staged.append(t)
            ptrs.append(t.data_ptr())
        arr_t = (C.c_void_p * len(ptrs))(*[C.c_void_p(p) if p else C.c_void_p(0) for p in ptrs])
        h = C.c_void_p()
        torch.cuda.synchronize(self.device)  # uploads done before the library's device-to-device copies
        self.ctx.check(self.L.vm_encoder_create(self.ctx.handle, C.byref(desc), arr_t, len(ptrs), C.byref(h)))
        torch.cuda.synchronize(self.device)
        del staged
        self.handle = h
        self.tokens = int(self.L.vm_encoder_tokens(h))
        self.out_dim = int(self.L.vm_encoder_out_dim(h))
        assert int(self.L.vm_encoder_patch_k(h)) == self.patch_k
        self.set_schedule(schedule)
        if micro_batch:
            self.set_option(_lib.VM_ENC_OPT_MICRO_BATCH, int(micro_batch))
        if last_layer != 3:
            self.set_option(_lib.VM_ENC_OPT_LAST_LAYER, int(last_layer))
        self._ws = None
        self._mean = (C.c_float * 3)(*spec["mean"])
        self._std = (C.c_float * 3)(*spec["std"])

    def set_option(self, option: int, value: int) -> None:
        self.ctx.check(self.L.vm_encoder_set_option(self.handle, int(option), int(value)))

    def set_schedule(self, schedule: str) -> None:
        """"auto" | "one_stream" | "two_streams" (vm_encoder_schedule)."""
        if schedule not in _lib.SCHEDULES:
            raise ValueError(f"schedule must be one of {sorted(_lib.SCHEDULES)}")
        self.set_option(_lib.VM_ENC_OPT_SCHEDULE, _lib.SCHEDULES[schedule])

    @property
    def schedule(self) -> str:
        v = int(self.L.vm_encoder_get_option(self.handle, _lib.VM_ENC_OPT_SCHEDULE))
        return {n: k for k, n in _lib.SCHEDULES.items()}[v]

    def close(self):
        if getattr(self, "handle", None):
            self.L.vm_encoder_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------------
    def preprocess(self, frames_u8: torch.Tensor, layout: str = "patches") -> torch.Tensor:
        """uint8 [B,H,W,3] BGR (device) -> [B, P, patch_k] (layout='patches') or [B,3,S,S] ('chw')."""
        if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[-1] != 3:
            raise ValueError("frames must be uint8 [B,H,W,3] (BGR, as cv2.VideoCapture.read returns them)")
        f = frames_u8.to(self.device).contiguous()
        B, Hh, Ww, _ = f.shape
        S, p = self.spec["image"], self.spec["patch"]
        if layout == "patches":
            out = torch.empty((B, (S // p) ** 2, self.patch_k), dtype=self.dtype, device=self.device)
            lay = _lib.VM_LAYOUT_PATCHES
        else:
            out = torch.empty((B, 3, S, S), dtype=self.dtype, device=self.device)
            lay = _lib.VM_LAYOUT_CHW
        self.ctx.check(self.L.vm_preprocess(
            self.ctx.handle, C.c_void_p(f.data_ptr()), B, Hh, Ww, self._mean, self._std, S,
            _lib.DTYPES[self.dtype_name], lay, p, self.patch_k, C.c_void_p(out.data_ptr()), _lib.current_stream_ptr()))
        f.record_stream(torch.cuda.current_stream())
        return out

    def patches_from_pixels(self, pixels_chw: torch.Tensor) -> torch.Tensor:
        """Test helper: already-normalised [B,3,S,S] pixels -> the patch-row layout (a pure permutation + pad)."""
        B, Cc, S, _ = pixels_chw.shape
        p = self.spec["patch"]
        g = S // p
        x = pixels_chw.reshape(B, Cc, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, Cc * p * p)
        x = torch.nn.functional.pad(x, (0, self.patch_k - x.shape[-1]))
        return x.to(device=self.device, dtype=self.dtype).contiguous()

    def workspace_bytes(self, B: int) -> int:
        return int(self.L.vm_encode_workspace_bytes(self.handle, int(B)))

    def micro_batch(self, B: int) -> int:
        """Frames vm_encode runs per pass when called with B frames (csrc/encoder.hip, micro_batch_of)."""
        return int(self.L.vm_encode_micro_batch(self.handle, int(B)))

    def new_workspace(self, B: int) -> torch.Tensor:
        """A workspace for batches of up to B frames that the CALLER owns (streaming sessions: a captured hipGraph
        bakes the address in, so it must not be the encoder's shared, growable one)."""
        return torch.empty(self.workspace_bytes(B), dtype=torch.uint8, device=self.device)

    def encode_patches(self, patches: torch.Tensor, l2_normalise: bool = True,
                       workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        if patches.dtype != self.dtype or patches.shape[1:] != (self.tokens - 1, self.patch_k):
            raise ValueError(f"patches must be [B,{self.tokens - 1},{self.patch_k}] {self.dtype}")
        patches = patches.contiguous()
        B = patches.shape[0]
        need = self.workspace_bytes(B)
        if workspace is None:
            # shared scratch of the eager path: calls are ordered on the current stream; a grown buffer replaces the
            # old one, which the caching allocator hands out again only behind this stream's pending work
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            workspace = self._ws
        elif workspace.numel() < need:
            raise ValueError(f"caller-owned encoder workspace {workspace.numel()} < {need} bytes")
        out = torch.empty((B, self.out_dim), dtype=self.dtype, device=self.device)
        self.ctx.check(self.L.vm_encode(self.handle, C.c_void_p(patches.data_ptr()), B, C.c_void_p(out.data_ptr()),
                                        1 if l2_normalise else 0, C.c_void_p(workspace.data_ptr()),
                                        workspace.numel(), _lib.current_stream_ptr()))
        patches.record_stream(torch.cuda.current_stream())
        return out

    def embed_frames(self, frames_u8: torch.Tensor, l2_normalise: bool = True,
                     workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.encode_patches(self.preprocess(frames_u8), l2_normalise, workspace)


class HipEmbedder:
    """Drop-in for the reference's embedder object: ``async aembed_query(x) -> List[float]`` where x is one frame
    (uint8 [H,W,3] BGR) instead of a caption string; ``aembed_frames`` embeds a whole chunk in one launch."""

    def __init__(self, encoder: FrameEncoder):
        self.encoder = encoder

    async def aembed_frames(self, frames) -> List[List[float]]:
        t = frames if isinstance(frames, torch.Tensor) else torch.as_tensor(np.asarray(frames))
        if t.dim() == 3:
            t = t.unsqueeze(0)
        emb = self.encoder.embed_frames(t)
        return emb.float().cpu().tolist()

    async def aembed_query(self, frame) -> List[float]:
        return (await self.aembed_frames(frame))[0]

    def embed_frames_tensor(self, frames: torch.Tensor) -> torch.Tensor:
        return self.encoder.embed_frames(frames)
