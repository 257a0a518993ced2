"""Streaming mode (BASELINE config C5): one hipGraph per chunk shape, replayed per incoming chunk.

The reference processes chunks serially and waits 3-35 s per chunk on a remote model
(src/pipeline/vlm_extractor.py:44-74).  Here the whole per-chunk hot path
    preprocess -> encode -> top-k over the rolling (ring) memory -> append
is captured ONCE into a HIP graph for a fixed chunk shape (B frames of HxW) and replayed; every kernel reads its row
count / ring cursor from device memory (csrc/memory.hip d_total), so replays see the memory grow and wrap.  No
allocation, no host sync and no launch-time host arguments change between replays - the property the C ABI promises.
"""
from __future__ import annotations

from typing import Tuple

import torch

from .encoder import FrameEncoder
from .memory import EmbeddingMemory


class StreamingSession:
    def __init__(self, encoder: FrameEncoder, memory: EmbeddingMemory, frames_per_chunk: int, height: int,
                 width: int, top_k: int = 10, warmup: int = 2):
        if not memory.ring:
            raise ValueError("streaming needs a ring memory (rolling window)")
        self.enc, self.mem, self.k = encoder, memory, top_k
        dev = encoder.device
        self.frames_in = torch.zeros((frames_per_chunk, height, width, 3), dtype=torch.uint8, device=dev)
        self.stream = torch.cuda.Stream(device=dev)
        self._appended_by_graph = 0
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):  # allocate workspaces / set kernel attributes outside the capture
                self._body()
            self.stream.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.emb, self.scores, self.rows = self._body()
        self._host_rows = len(memory)

    def _body(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        emb = self.enc.embed_frames(self.frames_in)
        scores, rows = self.mem.topk(emb, self.k, check_certified=False)   # uncertified queries are counted on device
        self.mem.append(emb)
        return emb, scores, rows

    def push(self, frames_u8: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Copy one chunk in, replay the graph; returns views of the graph's output buffers (valid until the next
        push).  The caller synchronises (or records an event) when it needs the values."""
        with torch.cuda.stream(self.stream):
            self.frames_in.copy_(frames_u8, non_blocking=True)
            self.graph.replay()
        self._host_rows += self.frames_in.shape[0]
        return self.emb, self.scores, self.rows

    @property
    def rows_appended(self) -> int:
        """Host-side count of rows pushed through the graph (the library's own host mirror does not advance under
        graph replay; the device-side counter does)."""
        return self._host_rows
