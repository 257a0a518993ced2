"""Hybrid-fusion wiring: the GPU vector leg inside the reference's hybrid retrieval, with the
candidate fusion and the ``true_chunks`` rank bookkeeping that its accuracy harness reads, unchanged.

``HybridRetriever._perform_hybrid_retrieval`` (src/pipeline/retriever_hybrid.py:142-282) calls, in this order,
``_vector_search_chunks`` (:167), ``_post_compress_chunks`` (:203) and then concatenates and truncates
(:241-244).  ``HipHybridMixin`` overrides the two methods with the HIP-backed ones (similarity.HipVectorSearch), so

    class GpuHybridRetriever(HipHybridMixin, HybridRetriever): ...
    r = GpuHybridRetriever(config, neo4j_handler, schedule_path); r.attach_memory(memory, min_score=0.3)

runs the reference's own control flow (entity search, graph hops, rerankers, formatting, ``batch-retrieve`` ->
``benchmark``) on the new vector path.  The pure functions restate the bookkeeping for callers that do not subclass.
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional, Sequence

from . import _lib
from .similarity import HipVectorSearch


def parse_chunk_index(chunk_id: Any) -> Optional[int]:
    """Last ``_``-separated field of a chunk id as int (retriever_hybrid.py:183-187, :253-257); None when it is not
    an integer.  Ids follow ``{run_uuid}_{batch_idx}_{i}`` (src/components/pre_llm_injector.py:91)."""
    try:
        return int(str(chunk_id).split("_")[-1])
    except Exception:
        return None


def true_chunk_rankings(chunks: Iterable[Dict[str, Any]], true_chunks: Sequence[int]) -> Dict[int, Optional[int]]:
    """1-based rank of the FIRST candidate whose parsed index equals each true chunk, None when absent
    (retriever_hybrid.py:173-190 for the vector hits, :247-262 for the final candidates)."""
    rankings: Dict[int, Optional[int]] = {idx: None for idx in true_chunks}
    for pos, c in enumerate(chunks, start=1):
        parsed = parse_chunk_index(c.get("id"))
        if parsed is not None and parsed in rankings and rankings[parsed] is None:
            rankings[parsed] = pos
    return rankings


def fuse_result_chunks(chunk_results: Optional[List[Dict]], traversal_chunks: Optional[List[Dict]],
                       top_k_chunks: int) -> List[Dict]:
    """Vector (or post-compressed / reranked) hits first, graph-traversal chunks behind them, cut to top_k_chunks;
    no de-duplication - the reference has none (retriever_hybrid.py:241-244)."""
    return ((chunk_results or []) + (traversal_chunks or []))[:top_k_chunks]


class HipHybridMixin:
    """Overrides the vector leg and the post-compression of ``HybridRetriever``; everything else is inherited.
    Expects the host class to provide ``self.config`` (RetrievalConfig) and ``self.embedder``."""

    _hip: Optional[HipVectorSearch] = None

    def attach_memory(self, memory, *, score_mode: int, min_score: float = 0.3, splitter=None):
        """``score_mode`` is required: see similarity.HipVectorSearch (Neo4j's score mapping is unpinned)."""
        self._hip = HipVectorSearch(memory, self.embedder, self.config, min_score=min_score, score_mode=score_mode,
                                    splitter=splitter)
        return self

    async def _vector_search_chunks(self, session, query):
        if self._hip is None:
            raise RuntimeError("attach_memory() first: there is no CPU / Cypher fallback on this path")
        return await self._hip._vector_search_chunks(session, query)

    async def _post_compress_chunks(self, query, chunks):
        if self._hip is None:
            raise RuntimeError("attach_memory() first: there is no CPU / Cypher fallback on this path")
        return await self._hip._post_compress_chunks(query, chunks)
