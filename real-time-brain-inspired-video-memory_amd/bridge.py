"""Neo4j stays the system of record: conversions between the device memory and the shapes in which the reference
moves ``Chunk.embedding`` around.

  * ``chunks_for_neo4j``      -> the ``text_chunks`` dicts ``{id, content, index, embedding}`` that
                                 ``Neo4jHandler._create_chunks_with_embeddings`` MERGEs (src/components/neo4j_handler.py:
                                 217-253; built at src/components/pre_llm_injector.py:88-97).
  * ``load_chunk_embeddings`` <- the ``Dict[chunk_id, List[float]]`` that ``PreLLMInjector._get_chunk_embeddings``
                                 returns (src/components/pre_llm_injector.py:390-412), same guards (:404-406).
  * ``export_nodes`` / ``load_export`` <-> the node list of ``GraphExporter.export_graph`` (format version "1.0",
                                 src/components/graph_exporter.py:60-66, node dicts :97-101: name / labels without
                                 "GraphNode" / properties without graph_uuid).

The memory stores 16-bit rows: values are rounded ONCE on the way in (``EmbeddingMemory._as_rows``) and every list
that leaves here holds the exact stored values as Python floats, so export -> load -> export is the identity.
The functions only need ``memory.append / id_of / meta_of / dim / dtype_name / graph_uuid / rows_host()``, which lets the
CPU tests drive them with a host stand-in; the native binary snapshot is ``EmbeddingMemory.snapshot / restore``.
"""
from __future__ import annotations

import json
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

EXPORT_FORMAT_VERSION = "1.0"  # src/components/graph_exporter.py:65


def rows_to_lists(rows_u16: np.ndarray, dtype: str) -> List[List[float]]:
    """Raw 16-bit patterns -> lists of Python floats (exact: fp16 / bf16 -> fp64 is lossless)."""
    if dtype == "f16":
        return rows_u16.view(np.float16).astype(np.float64).tolist()
    if dtype == "bf16":
        return (rows_u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64).tolist()
    raise ValueError(dtype)


def _row_span(memory, first_row: Optional[int], count: Optional[int]) -> Tuple[int, int, np.ndarray]:
    base, rows = memory.rows_host()
    lo = base if first_row is None else int(first_row)
    hi = base + rows.shape[0] if count is None else lo + int(count)
    if lo < base or hi > base + rows.shape[0]:
        raise IndexError(f"rows [{lo},{hi}) are not resident (memory holds [{base},{base + rows.shape[0]}))")
    return lo, hi, rows[lo - base:hi - base]


def chunks_for_neo4j(memory, first_row: Optional[int] = None, count: Optional[int] = None) -> List[Dict[str, Any]]:
    """Rows -> ``text_chunks`` for ``_create_chunks_with_embeddings``; ``index`` is the id's last ``_`` field when it
    is an integer (the reference's id scheme ``{run_uuid}_{batch_idx}_{i}``, pre_llm_injector.py:91), else None."""
    lo, hi, rows = _row_span(memory, first_row, count)
    out = []
    for r, emb in zip(range(lo, hi), rows_to_lists(rows, memory.dtype_name)):
        cid = memory.id_of(r)
        meta = memory.meta_of(r) or {}
        try:
            index = int(str(cid).split("_")[-1])
        except ValueError:
            index = None
        out.append({"id": cid, "content": meta.get("content"), "index": index, "embedding": emb})
    return out


def load_chunk_embeddings(memory, chunk_embeddings: Dict[str, Any]) -> Tuple[int, List[str]]:
    """Append what ``_get_chunk_embeddings`` returned, in dict order.  Entries the reference itself would not use are
    skipped and reported: non-list / empty embeddings and falsy ids (:404-406, and the truthiness filter at :362), and
    vectors whose length differs from the memory dimension (the reference scores those 0.0, :378-379)."""
    ids, rows, skipped = [], [], []
    for cid, emb in chunk_embeddings.items():
        if not cid or not isinstance(emb, list) or not emb or len(emb) != memory.dim:
            skipped.append(cid)
            continue
        ids.append(cid)
        rows.append(emb)
    if rows:
        memory.append(rows, ids=ids, meta=[None] * len(ids))
    return len(ids), skipped


def export_nodes(memory, embedding_model: Optional[str] = None, batch_id: Optional[int] = None
                 ) -> List[Dict[str, Any]]:
    """Resident rows as ``Chunk`` node dicts in the exporter's shape (graph_exporter.py:97-101)."""
    lo, hi, rows = _row_span(memory, None, None)
    nodes = []
    for r, emb in zip(range(lo, hi), rows_to_lists(rows, memory.dtype_name)):
        meta = memory.meta_of(r) or {}
        # property keys of the reference's own export (data/exports/mvp_...json, pinned by
        # tests/golden/export_excerpt.json): batch_id, created_at, id, content (+ embedding, embedding_model when the
        # chunk was stored with one, neo4j_handler.py:229-242)
        props = {"batch_id": meta.get("batch_id", batch_id), "created_at": meta.get("created_at"),
                 "id": memory.id_of(r), "content": meta.get("content"), "embedding": emb}
        if "time" in meta:
            props["time"] = meta["time"]
        if embedding_model is not None:
            props["embedding_model"] = embedding_model  # neo4j_handler.py:235
        nodes.append({"name": None, "labels": ["Chunk"], "properties": props})
    return nodes


def write_export(memory, path: str, **kw) -> str:
    import uuid
    # same keys, same order as GraphExporter.export_graph writes them (src/components/graph_exporter.py:61-67); the
    # reference fills export_timestamp with a fresh uuid4 string (:63, "Could use actual timestamp")
    data = {"graph_uuid": memory.graph_uuid, "export_timestamp": str(uuid.uuid4()),
            "nodes": export_nodes(memory, **kw), "relationships": [], "export_format_version": EXPORT_FORMAT_VERSION}
    with open(path, "w", encoding="utf-8") as f:
        json.dump(data, f, indent=2, ensure_ascii=False)
    return path


def load_export(memory, source, graph_uuid: Optional[str] = None) -> Tuple[int, List[Any]]:
    """Append every ``Chunk`` node with a usable embedding from an export file / dict, in file order.  Non-Chunk nodes
    and chunks stored without an embedding (neo4j_handler.py:243-253) are ignored; a ``graph_uuid`` mismatch raises."""
    data = source
    if isinstance(source, str):
        with open(source, "r", encoding="utf-8") as f:
            data = json.load(f)
    if str(data.get("export_format_version", EXPORT_FORMAT_VERSION)) != EXPORT_FORMAT_VERSION:
        raise ValueError(f"unsupported export_format_version {data.get('export_format_version')!r}")
    want = graph_uuid or memory.graph_uuid
    if want and data.get("graph_uuid") and data["graph_uuid"] != want:
        raise ValueError(f"export belongs to graph {data['graph_uuid']}, memory to {want}")
    ids, rows, metas, skipped = [], [], [], []
    for node in data.get("nodes", []):
        if "Chunk" not in node.get("labels", []):
            continue
        props = node.get("properties", {})
        emb, cid = props.get("embedding"), props.get("id")
        if not cid or not isinstance(emb, list) or len(emb) != memory.dim:
            if emb:
                skipped.append(cid)
            continue
        ids.append(cid)
        rows.append(emb)
        metas.append({"content": props.get("content"), "time": props.get("time"), "batch_id": props.get("batch_id"),
                      "created_at": props.get("created_at")})
    if rows:
        memory.append(rows, ids=ids, meta=metas)
    return len(ids), skipped
