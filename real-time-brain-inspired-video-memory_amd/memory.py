"""EmbeddingMemory: the HBM-resident replacement for the reference's embedding store.

Reference behaviour mirrored here:
  * append   = ``MERGE (c:Chunk:GraphNode {id}) SET c.embedding = $embedding``  (src/components/neo4j_handler.py:229-242);
               chunks whose embedding is falsy are stored WITHOUT one (:243-253) -> they never enter the search.
  * read-back = ``_get_chunk_embeddings`` (src/components/pre_llm_injector.py:390-412), which re-ships every stored
               vector over bolt for every batch; here rows stay on the device and only (row, score) pairs move.
  * row order = append order (the reference's dict order comes from an unordered Cypher MATCH and is not
               deterministic; the build defines it).

All arithmetic is in libvidmem.so (csrc/memory.hip, csrc/topk.hip, csrc/topk_exact.hip).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib


def _torch_dtype(name: str):
    return {"f16": torch.float16, "bf16": torch.bfloat16}[name]


class EmbeddingMemory:
    def __init__(self, capacity: int, dim: int, dtype: str = "f16", ring: bool = False, device: int = 0,
                 graph_uuid: Optional[str] = None):
        self.ctx = _lib.Context.get(device)
        self.L = self.ctx.L
        self.device = torch.device("cuda", device)
        self.dtype_name = dtype
        self.dtype = _torch_dtype(dtype)
        self.dim = int(dim)
        self.capacity = int(capacity)
        self.ring = bool(ring)
        self.graph_uuid = graph_uuid
        # Host tables: chunk id (reference pre_llm_injector.py:91) and {"time":..., "content":...} (for
        # _vector_search_chunks) of row (table_base + i).  table_base stays 0 unless a ring has wrapped far enough for
        # the slots of overwritten rows to be dropped (see _trim_tables); use id_of / meta_of for row -> entry.
        self.ids: List[Optional[str]] = []
        self.meta: List[Optional[dict]] = []
        self.table_base = 0
        h = C.c_void_p()
        self.ctx.check(self.L.vm_memory_create(self.ctx.handle, self.capacity, self.dim, _lib.DTYPES[dtype],
                                               1 if ring else 0, C.byref(h)))
        self.handle = h
        self._scratch = TopkScratch(self.device)

    def close(self):
        if getattr(self, "handle", None):
            self.L.vm_memory_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------------
    def __len__(self) -> int:
        return int(self.L.vm_memory_size(self.handle))

    @property
    def searchable(self) -> int:
        return min(len(self), self.capacity)

    def _as_rows(self, rows) -> torch.Tensor:
        """Accept a device/host tensor or a list of float lists; return a contiguous device tensor in the
        memory dtype (this is the quantisation point: the oracle is evaluated on these 16-bit values)."""
        if not isinstance(rows, torch.Tensor):
            rows = torch.tensor(rows, dtype=torch.float32)
        if rows.dim() == 1:
            rows = rows.unsqueeze(0)
        if rows.shape[-1] != self.dim:
            raise ValueError(f"embedding dimension {rows.shape[-1]} != memory dimension {self.dim}")
        return rows.to(device=self.device, dtype=self.dtype).contiguous()

    def append(self, rows, ids: Optional[Sequence[str]] = None, meta: Optional[Sequence[dict]] = None) -> int:
        t = self._as_rows(rows)
        B = t.shape[0]
        if ids is not None and len(ids) != B:
            raise ValueError("ids and rows differ in length")
        first = C.c_int64(0)
        self.ctx.check(self.L.vm_memory_append(self.handle, C.c_void_p(t.data_ptr()), B, C.byref(first),
                                               _lib.current_stream_ptr()))
        # the kernel reads `t` asynchronously: keep it alive until the stream has consumed it
        t.record_stream(torch.cuda.current_stream())
        self.ids.extend(list(ids) if ids is not None else [None] * B)
        self.meta.extend(list(meta) if meta is not None else [None] * B)
        self._trim_tables()
        return int(first.value)

    def _trim_tables(self) -> None:
        """A rolling window must not keep one table slot per row EVER appended: once a ring holds more than two
        capacities (+1024) of slots, those of rows that have been overwritten are dropped."""
        if self.ring and len(self.ids) > 2 * self.capacity + 1024:
            drop = len(self.ids) - self.capacity
            del self.ids[:drop], self.meta[:drop]
            self.table_base += drop

    def sync(self) -> int:
        """Bring the host mirror (row count, id / meta tables) in line with the device counter: call after hipGraph
        replays (streaming.StreamingSession) before eager appends, exhaustive searches or snapshots.  Rows appended
        by replays get ``None`` ids; a captured-but-never-run append is dropped."""
        total = int(self.L.vm_memory_sync(self.handle, _lib.current_stream_ptr()))
        if total < 0:
            self.ctx.check(total)
        n = max(0, total - self.table_base)
        del self.ids[n:], self.meta[n:]
        self.ids.extend([None] * (n - len(self.ids)))
        self.meta.extend([None] * (n - len(self.meta)))
        self._trim_tables()
        return total

    def prepare_topk(self, Q: int, k: int) -> None:
        """Size the shared top-k scratch for (Q, k) now, so a later eager call allocates nothing."""
        self._scratch = self._scratch.fit(self, Q, k)

    def reset(self):
        self.ctx.check(self.L.vm_memory_reset(self.handle, _lib.current_stream_ptr()))
        self.ids.clear()
        self.meta.clear()
        self.table_base = 0

    def rows_tensor(self) -> torch.Tensor:
        """Zero-copy view of the searchable physical rows [min(size, capacity), D] (debug / snapshot)."""
        n = self.searchable
        ptr = self.L.vm_memory_rows(self.handle)
        return _tensor_from_ptr(ptr, (n, self.dim), self.dtype, self.device)

    # ------------------------------------------------------------------------------------------------------
    def topk(self, queries, k: int, min_score: Optional[float] = None, score_mode: int = _lib.VM_SCORE_RAW,
             row_stride: int = 1, row_offset: int = 0, exact: bool = False, redo: bool = True,
             scratch: Optional["TopkScratch"] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (scores [Q,k] float64, rows [Q,k] int64, -1 padded), ordered (score desc, row asc): ALWAYS the
        exhaustive answer, as the reference returns it (src/components/pre_llm_injector.py:356-370).

        Fast path = fp32 MFMA scan + exact fp64 re-scoring (csrc/topk.hip), which marks the queries it cannot
        certify (more exact ties than candidate slots, gaps below the fp32 bound) in a per-query flag array;
        ``vm_topk_redo_flagged`` (csrc/topk_exact.hip) then redoes exactly those queries exhaustively on the device.
        Both read flags and row count from device memory: no host read-back, graph-capturable.  ``redo=False`` skips
        the second stage (measurement of the scan alone).  k > 58 and ``exact=True`` run the all-query exhaustive
        kernel.  A query of the wrong length scores 0.0 against every row in the reference (:378-379); here it raises
        in ``_as_rows`` unless the caller filters it (similarity.batch_similarities does).

        ``scratch``: workspaces + flag / counter buffers owned by the caller (streaming sessions capture their
        addresses into a hipGraph); default = this memory's own, used by eager calls on the current stream.
        """
        q = self._as_rows(queries)
        Q = q.shape[0]
        scores = torch.empty((Q, k), dtype=torch.float64, device=self.device)
        rows = torch.empty((Q, k), dtype=torch.int64, device=self.device)
        st = _lib.current_stream_ptr()
        use_min = 0 if min_score is None else 1
        ms = 0.0 if min_score is None else float(min_score)
        if not exact and k <= 58:
            if scratch is None:
                scratch = self._scratch = self._scratch.fit(self, Q, k)
            elif not scratch.fits(self, Q, k):
                raise ValueError("caller-owned top-k scratch is too small for this (Q, k)")
            self.ctx.check(self.L.vm_topk_cosine(
                self.handle, C.c_void_p(q.data_ptr()), Q, k, use_min, ms, int(score_mode), int(row_stride),
                int(row_offset), C.c_void_p(scores.data_ptr()), C.c_void_p(rows.data_ptr()),
                C.c_void_p(scratch.uncert.data_ptr()), C.c_void_p(scratch.flags.data_ptr()),
                C.c_void_p(scratch.ws.data_ptr()), scratch.ws.numel(), st))
            if redo:
                self.ctx.check(self.L.vm_topk_redo_flagged(
                    self.handle, C.c_void_p(q.data_ptr()), Q, k, use_min, ms, int(score_mode), int(row_stride),
                    int(row_offset), C.c_void_p(scratch.flags.data_ptr()), C.c_void_p(scores.data_ptr()),
                    C.c_void_p(rows.data_ptr()), C.c_void_p(scratch.redo_ws.data_ptr()), scratch.redo_ws.numel(),
                    st))
            q.record_stream(torch.cuda.current_stream())
            return scores, rows
        need = int(self.L.vm_topk_exact_workspace_bytes(self.handle, Q, k))
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
        self.ctx.check(self.L.vm_topk_cosine_exact(
            self.handle, C.c_void_p(q.data_ptr()), Q, k, use_min, ms, int(score_mode), int(row_stride),
            int(row_offset), C.c_void_p(scores.data_ptr()), C.c_void_p(rows.data_ptr()),
            C.c_void_p(ws.data_ptr()), ws.numel(), st))
        ws.record_stream(torch.cuda.current_stream())
        return scores, rows

    @property
    def uncertified_count(self) -> int:
        """Queries redone exhaustively since ``reset_uncertified`` on this memory's own scratch (one 4-byte
        read-back; synchronises)."""
        return int(self._scratch.uncert.item())

    def reset_uncertified(self) -> None:
        self._scratch.uncert.zero_()

    def cosine_exact(self, queries, rows, as_f32: bool = False) -> torch.Tensor:
        """All-pairs reference cosine [Q,S] float64 between two row sets (neither needs to be stored).
        ``as_f32``: score the vectors as fp32 values instead of rounding them to the memory's 16-bit type first - what
        the post-compression filter uses, because its operands are fresh embedder outputs that never enter the
        memory and the reference compares the un-rounded floats with the threshold (retriever_hybrid.py:497-499)."""
        if as_f32:
            def f32(x):
                t = x if isinstance(x, torch.Tensor) else torch.tensor(x, dtype=torch.float32)
                if t.dim() == 1:
                    t = t.unsqueeze(0)
                return t.to(device=self.device, dtype=torch.float32).contiguous()
            q, r = f32(queries), f32(rows)
            if q.shape[-1] != r.shape[-1]:
                raise ValueError(f"embedding dimensions differ: {q.shape[-1]} vs {r.shape[-1]}")
            dim, dt = q.shape[-1], _lib.VM_F32
        else:
            q, r = self._as_rows(queries), self._as_rows(rows)
            dim, dt = self.dim, _lib.DTYPES[self.dtype_name]
        out = torch.empty((q.shape[0], r.shape[0]), dtype=torch.float64, device=self.device)
        self.ctx.check(self.L.vm_cosine_exact(self.ctx.handle, C.c_void_p(q.data_ptr()), q.shape[0],
                                              C.c_void_p(r.data_ptr()), r.shape[0], dim, dt,
                                              C.c_void_p(out.data_ptr()), _lib.current_stream_ptr()))
        q.record_stream(torch.cuda.current_stream())
        r.record_stream(torch.cuda.current_stream())
        return out

    # ---- persistence: the reference's only durable store is the `embedding` list property
    # (src/components/neo4j_handler.py:229-242) and the export JSON (src/components/graph_exporter.py:61-67);
    # here: raw 16-bit rows in row-id order + the host id / meta tables, one .npz ------------------------------
    def rows_host(self):
        """(first_row_id, uint16 [n, D]): the searchable rows as raw 16-bit patterns in row-id order (host copy)."""
        import numpy as np
        total, n = len(self), self.searchable
        phys = self.rows_tensor().view(torch.int16)
        if self.ring and total > self.capacity:  # oldest row sits at slot total % capacity
            head = total % self.capacity
            phys = torch.cat([phys[head:], phys[:head]])
        return total - n, phys.cpu().numpy().view(np.uint16)

    def snapshot(self, path: str) -> None:
        import json
        import numpy as np
        base, rows = self.rows_host()
        total = base + rows.shape[0]
        np.savez(path, rows=rows, dtype=self.dtype_name, dim=self.dim,
                 first_row_id=base, graph_uuid=self.graph_uuid or "",
                 ids=json.dumps([self.id_of(r) for r in range(base, total)]),
                 meta=json.dumps([self.meta_of(r) for r in range(base, total)]))

    @classmethod
    def restore(cls, path: str, capacity: Optional[int] = None, ring: bool = False, device: int = 0
                ) -> "EmbeddingMemory":
        """Row ids restart at 0 in the restored memory (ids / meta tables are restored in the same order)."""
        import json
        import numpy as np
        z = np.load(path, allow_pickle=False)
        rows = torch.from_numpy(z["rows"].view(np.int16))
        dtype = str(z["dtype"])
        mem = cls(capacity or max(rows.shape[0], 1), int(z["dim"]), dtype, ring=ring, device=device,
                  graph_uuid=str(z["graph_uuid"]) or None)
        if rows.shape[0]:
            mem.append(rows.view(_torch_dtype(dtype)), ids=json.loads(str(z["ids"])), meta=json.loads(str(z["meta"])))
        return mem

    def id_of(self, row: int) -> Optional[str]:
        i = row - self.table_base
        return self.ids[i] if 0 <= i < len(self.ids) else None

    def meta_of(self, row: int) -> Optional[dict]:
        i = row - self.table_base
        return self.meta[i] if 0 <= i < len(self.meta) else None


class TopkScratch:
    """Device buffers one stream of top-k calls needs: the scan workspace, the redo workspace, the per-query flag
    array and the uncertified counter.  An owner keeps ONE instance per stream / captured graph: a hipGraph bakes the
    addresses in, so a scratch a capture has seen is never re-allocated (``fit`` returns a NEW object when it has to
    grow and leaves the old buffers to whoever still references them)."""

    def __init__(self, device, Q: int = 0, k: int = 0, ws_bytes: int = 0, redo_bytes: int = 0):
        self.Q, self.k = int(Q), int(k)
        self.ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=device)
        self.redo_ws = torch.empty(max(redo_bytes, 256), dtype=torch.uint8, device=device)
        self.flags = torch.zeros(max(Q, 1), dtype=torch.int32, device=device)
        self.uncert = torch.zeros(1, dtype=torch.int32, device=device)

    @staticmethod
    def _need(memory: "EmbeddingMemory", Q: int, k: int):
        return (int(memory.L.vm_topk_workspace_bytes(memory.handle, int(Q), int(k))),
                int(memory.L.vm_topk_redo_workspace_bytes(memory.handle, int(Q), int(k))))

    def fits(self, memory: "EmbeddingMemory", Q: int, k: int) -> bool:
        ws, redo = self._need(memory, Q, k)
        return self.ws.numel() >= ws and self.redo_ws.numel() >= redo and self.flags.numel() >= Q

    def fit(self, memory: "EmbeddingMemory", Q: int, k: int) -> "TopkScratch":
        if self.fits(memory, Q, k):
            return self
        ws, redo = self._need(memory, Q, k)
        new = TopkScratch(self.ws.device, max(Q, self.flags.numel()), k, max(ws, self.ws.numel()),
                          max(redo, self.redo_ws.numel()))
        new.uncert.copy_(self.uncert)
        return new

    @classmethod
    def for_(cls, memory: "EmbeddingMemory", Q: int, k: int) -> "TopkScratch":
        ws, redo = cls._need(memory, Q, k)
        return cls(memory.device, Q, k, ws, redo)


def topk_select(ctx: "_lib.Context", scores: torch.Tensor, k: int, col_limit: Optional[torch.Tensor] = None,
                row_base: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """[Q,S] fp64 all-pairs scores -> the k best columns of every row by (score descending, column ascending), query q
    ranking only columns < col_limit[q] (vm_topk_select; csrc/topk_exact.hip).  Rows = row_base + column, -1 padded."""
    Q, S = scores.shape
    scores = scores.contiguous()
    lim = None
    if col_limit is not None:
        lim = col_limit.to(device=scores.device, dtype=torch.int64).contiguous()
        if lim.shape != (Q,):
            raise ValueError("col_limit must hold one limit per query")
    out_s = torch.empty((Q, k), dtype=torch.float64, device=scores.device)
    out_r = torch.empty((Q, k), dtype=torch.int64, device=scores.device)
    ctx.check(ctx.L.vm_topk_select(ctx.handle, C.c_void_p(scores.data_ptr()), Q, S,
                                   C.c_void_p(lim.data_ptr() if lim is not None else 0), int(k), int(row_base),
                                   C.c_void_p(out_s.data_ptr()), C.c_void_p(out_r.data_ptr()),
                                   _lib.current_stream_ptr()))
    return out_s, out_r


def topk_merge(ctx: "_lib.Context", scores: torch.Tensor, rows: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[parts,Q,k] per-shard results -> global [Q,k] (csrc/topk.hip topk_merge_kernel)."""
    parts, Q, k = scores.shape
    scores = scores.contiguous()
    rows = rows.contiguous()
    out_s = torch.empty((Q, k), dtype=torch.float64, device=scores.device)
    out_r = torch.empty((Q, k), dtype=torch.int64, device=scores.device)
    ctx.check(ctx.L.vm_topk_merge(ctx.handle, C.c_void_p(scores.data_ptr()), C.c_void_p(rows.data_ptr()), parts, Q,
                                  k, C.c_void_p(out_s.data_ptr()), C.c_void_p(out_r.data_ptr()),
                                  _lib.current_stream_ptr()))
    return out_s, out_r


def _tensor_from_ptr(ptr: int, shape, dtype, device) -> torch.Tensor:
    """Wrap a raw device pointer owned by libvidmem as a torch tensor (no copy, no ownership)."""
    import numpy as np

    class _Holder:
        pass

    n = 1
    for s in shape:
        n *= s
    itemsize = torch.empty(0, dtype=dtype).element_size()
    h = _Holder()
    h.__cuda_array_interface__ = {
        "shape": (n * itemsize,), "typestr": "|u1", "data": (int(ptr), False), "version": 3, "strides": None,
    }
    raw = torch.as_tensor(h, device=device)
    return raw.view(dtype).view(*shape)
