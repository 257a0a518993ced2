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
        self._ws = None
        self._uncert = torch.zeros(1, dtype=torch.int32, device=self.device)

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
        """Size the top-k workspace for (Q, k) now, so a later graph capture allocates nothing."""
        self._workspace(int(self.L.vm_topk_workspace_bytes(self.handle, int(Q), int(k))))

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
    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=self.device)
        return self._ws

    def topk(self, queries, k: int, min_score: Optional[float] = None, score_mode: int = _lib.VM_SCORE_RAW,
             row_stride: int = 1, row_offset: int = 0, exact: bool = False,
             check_certified: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (scores [Q,k] float64, rows [Q,k] int64, -1 padded), ordered (score desc, row asc).

        Fast path = fp32 MFMA scan + exact fp64 re-scoring (csrc/topk.hip).  Queries the scan cannot certify, and
        k > 58, go through the exhaustive fp64 kernel (csrc/topk_exact.hip).  Both are GPU paths.
        """
        q = self._as_rows(queries)
        Q = q.shape[0]
        scores = torch.empty((Q, k), dtype=torch.float64, device=self.device)
        rows = torch.empty((Q, k), dtype=torch.int64, device=self.device)
        st = _lib.current_stream_ptr()
        use_min = 0 if min_score is None else 1
        ms = 0.0 if min_score is None else float(min_score)
        if not exact and k <= 58:
            need = int(self.L.vm_topk_workspace_bytes(self.handle, Q, k))
            ws = self._workspace(need)
            if check_certified:
                self._uncert.zero_()  # otherwise the counter keeps accumulating (read it with .item() later)
            self.ctx.check(self.L.vm_topk_cosine(
                self.handle, C.c_void_p(q.data_ptr()), Q, k, use_min, ms, int(score_mode), int(row_stride),
                int(row_offset), C.c_void_p(scores.data_ptr()), C.c_void_p(rows.data_ptr()),
                C.c_void_p(self._uncert.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), st))
            self.last_uncertified = None
            if not check_certified:
                return scores, rows
            self.last_uncertified = int(self._uncert.item())  # one 4-byte readback; results are read next anyway
            if self.last_uncertified == 0:
                return scores, rows
        need = int(self.L.vm_topk_exact_workspace_bytes(self.handle, Q, k))
        ws = self._workspace(need)
        self.ctx.check(self.L.vm_topk_cosine_exact(
            self.handle, C.c_void_p(q.data_ptr()), Q, k, use_min, ms, int(score_mode), int(row_stride),
            int(row_offset), C.c_void_p(scores.data_ptr()), C.c_void_p(rows.data_ptr()),
            C.c_void_p(ws.data_ptr()), ws.numel(), st))
        return scores, rows

    def cosine_exact(self, queries, rows) -> torch.Tensor:
        """All-pairs reference cosine [Q,S] float64 between two row sets (neither needs to be stored)."""
        q = self._as_rows(queries)
        r = self._as_rows(rows)
        out = torch.empty((q.shape[0], r.shape[0]), dtype=torch.float64, device=self.device)
        self.ctx.check(self.L.vm_cosine_exact(self.ctx.handle, C.c_void_p(q.data_ptr()), q.shape[0],
                                              C.c_void_p(r.data_ptr()), r.shape[0], self.dim,
                                              _lib.DTYPES[self.dtype_name], C.c_void_p(out.data_ptr()),
                                              _lib.current_stream_ptr()))
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
