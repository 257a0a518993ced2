"""FrameEmbeddingExtractor: drop-in for the reference's VLMExtractor on the frame-embedding path.

Mirrors ``VLMExtractor`` (src/pipeline/vlm_extractor.py):
  * ctor takes the pipeline config (only ``config.video.chunk_size_seconds`` / ``frames_per_chunk`` are read, :38,:101);
  * ``async process_video(video_path, output_path) -> str`` runs the same serial chunk loop (:44-74) with the same
    integers: chunk_size_frames = int(chunk_size_seconds * fps) (:38), total_chunks = max(1, int(total/chunk)) (:39),
    frame picks np.linspace(start, end-1, n, dtype=int) (:107), downscale to 720 rows when taller (:114-116), label
    "MM:SS-MM:SS" (:57-59), and writes the same JSON (:77-88) - ``content`` carries a short placeholder and the new
    fields ``embedding_rows`` / ``similar`` carry what the hot path produced for the chunk.
  * where the reference POSTs JPEGs to a remote VLM (:121-185), this encodes the frames on the GPU, retrieves the top-k
    most similar stored frames and appends the new embeddings to the memory.

Frame sources: an OpenCV-readable video when ``cv2`` is importable (it is NOT installed in the build image), or an
``.npy`` / ``.npz`` stack of uint8 BGR frames ``[N,H,W,3]`` (``.npz``: arrays ``frames`` and optional ``fps``).
Frame decode is the next row of the scope table, not part of this round's GPU path.
"""
from __future__ import annotations

import json
import time
import uuid
from typing import Any, List, Optional, Tuple

import numpy as np
import torch

from .encoder import FrameEncoder
from .memory import EmbeddingMemory


class _ArraySource:
    def __init__(self, frames: np.ndarray, fps: float):
        self.frames, self.fps, self.total = frames, float(fps), int(frames.shape[0])

    def read(self, idx: int) -> Optional[np.ndarray]:
        return self.frames[idx] if 0 <= idx < self.total else None

    def release(self):
        pass


class _Cv2Source:
    def __init__(self, path: str):
        import cv2  # noqa: F401  (absent in the build image; present on a deployment box)
        self.cv2 = cv2
        self.cap = cv2.VideoCapture(path)
        self.fps = self.cap.get(cv2.CAP_PROP_FPS)
        self.total = int(self.cap.get(cv2.CAP_PROP_FRAME_COUNT))

    def read(self, idx: int) -> Optional[np.ndarray]:
        self.cap.set(self.cv2.CAP_PROP_POS_FRAMES, idx)
        ret, frame = self.cap.read()
        return frame if ret else None

    def release(self):
        self.cap.release()


def open_source(video_path: str, default_fps: float = 30.0):
    if video_path.endswith(".npy"):
        return _ArraySource(np.load(video_path, mmap_mode="r"), default_fps)
    if video_path.endswith(".npz"):
        z = np.load(video_path)
        return _ArraySource(z["frames"], float(z["fps"]) if "fps" in z else default_fps)
    return _Cv2Source(video_path)


def chunk_plan(fps: float, total_frames: int, chunk_size_seconds: float, frames_per_chunk: int):
    """The reference's chunking integers (src/pipeline/vlm_extractor.py:38-46, :101-107, :57-59)."""
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
        start_time, end_time = start_frame / fps, end_frame / fps
        time_str = (f"{int(start_time // 60):02d}:{int(start_time % 60):02d}-"
                    f"{int(end_time // 60):02d}:{int(end_time % 60):02d}")
        plan.append((chunk_idx, start_frame, end_frame, [int(i) for i in idx], time_str))
    return total_chunks, plan


class FrameEmbeddingExtractor:
    def __init__(self, config: Any, encoder: FrameEncoder, memory: EmbeddingMemory, top_k: int = 5,
                 stager_factory=None):
        self.config = config
        self.encoder = encoder
        self.memory = memory
        self.top_k = top_k
        self.timings: List[Tuple[str, float]] = []
        self._stager = None
        self._stager_factory = stager_factory  # (frames, H, W, device) -> ingest.FrameStager-like; tests inject a host one

    def _read_chunk(self, src, indices) -> List[np.ndarray]:
        """The reference's per-chunk frame pick (:107-111): unreadable frames are dropped.  Frames taller than 720
        rows are NOT downscaled on the host: the preprocess kernel's own bilinear resize takes them to the encoder
        input in one pass (the reference's cv2.resize to 720 rows, :114-116, only shrinks the HTTP payload)."""
        return [f for f in (src.read(i) for i in indices) if f is not None]

    def _stage(self, frames: List[np.ndarray]):
        """Start the H2D copy of a chunk (ingest.FrameStager: pinned slot + copy stream); the stager is sized on the
        first chunk and rebuilt if the frame size changes."""
        h, w = frames[0].shape[:2]
        cap = max(len(frames), int(self.config.video.frames_per_chunk))
        if self._stager is None or self._stager.shape[1:3] != (h, w) or self._stager.shape[0] < len(frames):
            factory = self._stager_factory
            if factory is None:
                from .ingest import FrameStager as factory
            self._stager = factory(cap, h, w, self.encoder.device.index or 0)
        return self._stager.stage(frames)

    async def process_video(self, video_path: str, output_path: str) -> str:
        run_id = str(uuid.uuid4())
        src = open_source(video_path)
        try:
            fps, total_frames = src.fps, src.total
            total_chunks, plan = chunk_plan(fps, total_frames, self.config.video.chunk_size_seconds,
                                            self.config.video.frames_per_chunk)
            results = []

            def read_and_stage(pi):
                """Host side of chunk pi: frame reads + pinned staging + start of the H2D copy (None: no frames)."""
                if pi >= len(plan):
                    return None
                frames = self._read_chunk(src, plan[pi][3])
                return (len(frames), self._stage(frames)) if frames else None

            staged = read_and_stage(0)
            for pi, (chunk_idx, start, end, indices, time_str) in enumerate(plan):
                chunk_start = time.perf_counter()
                current, emb = staged, None
                if current is not None:
                    nframes, ticket = current
                    emb = self.encoder.embed_frames(self._stager.get(ticket))   # asynchronous launches
                    self._stager.done(ticket)
                # while the GPU encodes chunk pi, the host reads chunk pi+1 and its frames cross PCIe
                staged = read_and_stage(pi + 1)
                if emb is None:
                    continue
                similar = []
                if self.memory.searchable and self.top_k > 0:
                    scores, rows = self.memory.topk(emb, self.top_k)
                    for s_row, r_row in zip(scores.cpu().tolist(), rows.cpu().tolist()):
                        similar.append([(self.memory.id_of(r), float(s)) for s, r in zip(s_row, r_row) if r >= 0])
                ids = [f"{run_id}_{chunk_idx}_{i}" for i in range(nframes)]  # pre_llm_injector.py:91 id scheme
                first = self.memory.append(emb, ids=ids, meta=[{"time": time_str, "content": None}] * nframes)
                torch.cuda.synchronize(self.encoder.device)
                chunk_time = time.perf_counter() - chunk_start
                results.append({
                    "time": time_str,
                    "content": f"[{nframes} frame embeddings]",
                    "chunk_idx": chunk_idx,
                    "processing_time": chunk_time,
                    "embedding_rows": list(range(first, first + nframes)),
                    "similar": similar,
                })
                self.timings.append((f"chunk_{chunk_idx}", chunk_time))
            cfg = self.config.dict() if hasattr(self.config, "dict") else None
            output_data = {
                "metadata": {"run_id": run_id, "video_path": video_path, "total_chunks": total_chunks, "config": cfg},
                "results": results,
            }
            with open(output_path, "w") as f:
                json.dump(output_data, f, indent=2)
            return output_path
        finally:
            src.release()
