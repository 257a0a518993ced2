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
from .memory import EmbeddingMemory, TopkScratch


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
        self._stager = None
        # Everything the captured graph writes through lives in buffers THIS session owns for its whole life: the
        # encoder workspace, the top-k workspaces, the per-query flags and the uncertified counter.  The encoder's and
        # the memory's own scratch may be re-allocated by a later, larger eager call, and eager calls on another
        # stream would race a replay on shared scratch.
        self._enc_ws = encoder.new_workspace(frames_per_chunk)
        self._scratch = TopkScratch.for_(memory, frames_per_chunk, top_k)
        with torch.cuda.stream(self.stream):
            # Warm up (code objects, kernel attributes, encoder workspace) against a scratch ring so that the user's
            # memory is not touched: a graph capture records launches without running them.
            scratch = EmbeddingMemory(max(4 * frames_per_chunk, 64), memory.dim, memory.dtype_name, ring=True,
                                      device=dev.index or 0)
            scratch.append(torch.zeros((1, memory.dim), dtype=memory.dtype, device=dev))
            for _ in range(max(1, warmup)):
                self._body(scratch)
            scratch.close()
            self.stream.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.emb, self.scores, self.rows = self._body(memory)
            memory.sync()   # the capture advanced the host-side row count without appending anything: undo that
        self._host_rows = len(memory)

    def _body(self, mem: EmbeddingMemory) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        emb = self.enc.embed_frames(self.frames_in, workspace=self._enc_ws)
        # exhaustive answers inside the graph: the scan flags what it cannot certify, the redo kernel reads the flags
        # on the device.  The counter is zeroed per replay, so after a push it is THAT push's number of redone queries.
        self._scratch.uncert.zero_()
        scores, rows = mem.topk(emb, self.k, scratch=self._scratch)
        mem.append(emb)
        return emb, scores, rows

    @property
    def uncertified_last_push(self) -> int:
        """Queries of the most recent push that the fp32 scan could not certify and the exhaustive kernel redid
        inside the same replay (results are exhaustive either way).  Synchronises the session stream."""
        self.stream.synchronize()
        return int(self._scratch.uncert.item())

    def push(self, frames_u8: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Copy one chunk in, replay the graph; returns views of the graph's output buffers (valid until the next
        push).  The caller synchronises (or records an event) when it needs the values."""
        with torch.cuda.stream(self.stream):
            self.frames_in.copy_(frames_u8, non_blocking=True)
            self.graph.replay()
        self._host_rows += self.frames_in.shape[0]
        return self.emb, self.scores, self.rows

    # ---- host frames: pinned slots + copy stream, see ingest.FrameStager -------------------------
    @property
    def stager(self):
        if self._stager is None:
            from .ingest import FrameStager
            F, H, W, _ = self.frames_in.shape
            self._stager = FrameStager(F, H, W, device=self.enc.device.index or 0, depth=2)
        return self._stager

    def push_staged(self, ticket) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Replay on a chunk whose H2D copy was started earlier (``stager.stage`` / ``next_slot`` + ``commit``); only
        full chunks (the graph's shape).  The session stream waits for the copy, the host does not."""
        if ticket.count != self.frames_in.shape[0]:
            raise ValueError("the captured graph processes full chunks only")
        with torch.cuda.stream(self.stream):
            staged = self.stager.get(ticket)
            self.frames_in.copy_(staged, non_blocking=True)   # D2D into the graph's input buffer (~80 us at 1080p)
            self.stager.done(ticket)
            self.graph.replay()
        self._host_rows += self.frames_in.shape[0]
        return self.emb, self.scores, self.rows

    def push_host(self, frames) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Host frames (uint8 BGR [F,H,W,3] array or list): stage + replay.  To overlap the next chunk's PCIe copy
        with this replay, call ``stager.stage(next_frames)`` right after and ``push_staged`` it later."""
        return self.push_staged(self.stager.stage(frames))

    def sync(self) -> int:
        """Wait for the replays issued so far and bring the memory's host-side tables in line (EmbeddingMemory.sync):
        call before eager appends / exhaustive searches / snapshots on the same memory."""
        with torch.cuda.stream(self.stream):
            return self.mem.sync()

    @property
    def rows_appended(self) -> int:
        """Host-side count of rows in the memory including those pushed through the graph (the library's own host
        mirror does not advance under graph replay; the device-side counter does, see ``sync``)."""
        return self._host_rows
