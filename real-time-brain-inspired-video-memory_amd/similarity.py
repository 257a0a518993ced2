"""Drop-in replacements for the reference's similarity call sites, backed by EmbeddingMemory (HIP).

Same names, argument meaning, return shapes and error convention as the reference:

  * ``HipPreLLMSimilarity._calculate_batch_similarities(chunk_embeddings, neo4j_handler)``
        <- PreLLMInjector._calculate_batch_similarities, src/components/pre_llm_injector.py:346-372
        returns List[Q] of List[<=k] of (chunk_id, score); an Exception entry yields [] (:357-359);
        k = embedder_config.top_k_chunk_with_batch_similarity.
  * ``merge_batch_similarities``          <- the max-merge of src/components/pre_llm_injector.py:238-249
        (host-side dict logic of the CALLER, kept as the reference has it).
  * ``HipVectorSearch._vector_search_chunks(session, query)``
        <- HybridRetriever._vector_search_chunks, src/pipeline/retriever_hybrid.py:284-323
        returns List[{id,time,content,score,source:"vector"}]; any failure -> [] (:321-323).
  * ``HipVectorSearch._post_compress_chunks(query, chunks)``
        <- HybridRetriever._post_compress_chunks, src/pipeline/retriever_hybrid.py:465-514
        keeps segments with cosine >= compression_threshold in encounter order, then [:top_k];
        any failure -> chunks unchanged (:512-514).

Hot-path helpers never raise (they log and degrade), exactly like the reference; programmer errors
(shape / dtype, VidmemError VM_ERR_INVALID) do.
"""
from __future__ import annotations

import logging
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from .memory import EmbeddingMemory

logger = logging.getLogger("vidmem.similarity")


def merge_batch_similarities(batch_similarities: Sequence[Sequence[Tuple[str, float]]],
                             top_k_similar_batch: int) -> List[Tuple[str, float]]:
    """src/components/pre_llm_injector.py:238-249: max score per chunk id (first-seen order), stable descending
    sort, first ``top_k_similar_batch``."""
    final_scores: Dict[str, float] = {}
    for chunk_similarities in batch_similarities:
        for chunk_id, score in chunk_similarities:
            if chunk_id not in final_scores or score > final_scores[chunk_id]:
                final_scores[chunk_id] = score
    final_score_list = sorted(final_scores.items(), key=lambda x: x[1], reverse=True)
    return final_score_list[:top_k_similar_batch]


def _length(e) -> int:
    return int(e.shape[-1]) if isinstance(e, torch.Tensor) else len(e)


def batch_similarities(memory: EmbeddingMemory, chunk_embeddings: Sequence, top_k: int
                       ) -> List[List[Tuple[str, float]]]:
    """One batched top-k launch for every non-failed query; result re-threaded into the reference's list shape.

    A query whose length differs from the stored vectors' scores 0.0 against EVERY row in the reference
    (``_cosine_similarity``: ``if len(vec1) != len(vec2): return 0.0``, src/components/pre_llm_injector.py:378-379), and
    the stable descending sort (:369) then keeps memory order: the answer is the first ``top_k`` stored chunks, each
    with score 0.0.  That is reproduced on the host (no arithmetic involved) instead of raising."""
    ok_idx = [i for i, e in enumerate(chunk_embeddings) if not isinstance(e, Exception) and e is not None]
    out: List[List[Tuple[str, float]]] = [[] for _ in chunk_embeddings]
    if not ok_idx or memory.searchable == 0 or top_k <= 0:
        return out
    wrong = [i for i in ok_idx if _length(chunk_embeddings[i]) != memory.dim]
    if wrong:
        first_row = len(memory) - memory.searchable
        zeros = [(memory.id_of(first_row + j), 0.0) for j in range(min(top_k, memory.searchable))]
        for i in wrong:
            out[i] = list(zeros)
        ok_idx = [i for i in ok_idx if i not in set(wrong)]
        if not ok_idx:
            return out
    first = chunk_embeddings[ok_idx[0]]
    if isinstance(first, torch.Tensor):
        q = torch.stack([chunk_embeddings[i] for i in ok_idx])
    else:
        q = torch.tensor([list(chunk_embeddings[i]) for i in ok_idx], dtype=torch.float32)
    scores, rows = memory.topk(q, top_k)
    scores, rows = scores.cpu().tolist(), rows.cpu().tolist()
    for slot, i in enumerate(ok_idx):
        out[i] = [(memory.id_of(r), float(s)) for r, s in zip(rows[slot], scores[slot]) if r >= 0]
    return out


class HipPreLLMSimilarity:
    """Mixin / stand-alone object for PreLLMInjector: set ``self.memory`` and ``self.embedder_config``."""

    def __init__(self, memory: EmbeddingMemory, embedder_config: Any):
        self.memory = memory
        self.embedder_config = embedder_config

    async def _calculate_batch_similarities(self, chunk_embeddings, neo4j_handler=None
                                            ) -> List[List[Tuple[str, float]]]:
        try:
            return batch_similarities(self.memory, chunk_embeddings,
                                      self.embedder_config.top_k_chunk_with_batch_similarity)
        except _lib.VidmemError as e:
            if e.code == _lib.VM_ERR_INVALID:
                raise
            logger.warning("similarity search failed: %s", e)  # reference: log and degrade to empty
            return [[] for _ in chunk_embeddings]


def _zip_truncating_cosine(vec1, vec2) -> float:
    """src/pipeline/retriever_hybrid.py:655-664, for operands of different lengths only (equal lengths go to the device)."""
    import math
    dot_product = sum(a * b for a, b in zip(vec1, vec2))
    mag1 = math.sqrt(sum(a * a for a in vec1))
    mag2 = math.sqrt(sum(b * b for b in vec2))
    if mag1 * mag2 == 0:
        return 0.0
    return dot_product / (mag1 * mag2)


class HipVectorSearch:
    """Mixin / stand-alone object for HybridRetriever's vector leg."""

    def __init__(self, memory: EmbeddingMemory, embedder: Any, config: Any, *, score_mode: int,
                 min_score: float = 0.3, splitter: Optional[Callable[[str], List[str]]] = None):
        """``score_mode`` is REQUIRED (keyword): the reference filters on Neo4j's
        ``vector.similarity.cosine(...) > 0.3`` (src/pipeline/retriever_hybrid.py:296-298), a third-party function of an
        unpinned server image whose value may be the raw cosine or its [0,1] mapping (1+cos)/2 - with the literal 0.3
        meaning cos > 0.3 in one case and cos > -0.4 in the other.  Nothing in the reference pins it (parity unpinned,
        SURVEY.md 8 a10), so the integrator states which one their deployment had: ``_lib.VM_SCORE_RAW`` or
        ``_lib.VM_SCORE_UNIT_INTERVAL``; ``min_score`` (default: the reference's literal) is compared AFTER the mapping."""
        if score_mode not in (_lib.VM_SCORE_RAW, _lib.VM_SCORE_UNIT_INTERVAL):
            raise ValueError("score_mode must be VM_SCORE_RAW or VM_SCORE_UNIT_INTERVAL")
        self.memory, self.embedder, self.config = memory, embedder, config
        self.min_score, self.score_mode, self.splitter = min_score, score_mode, splitter

    async def _vector_search_chunks(self, session, query) -> List[Dict[str, Any]]:
        try:
            query_embedding = await self.embedder.aembed_query(query)
            scores, rows = self.memory.topk([query_embedding], self.config.top_k_chunks, min_score=self.min_score,
                                            score_mode=self.score_mode)
            chunks = []
            for r, s in zip(rows[0].cpu().tolist(), scores[0].cpu().tolist()):
                if r < 0:
                    continue
                meta = self.memory.meta_of(r) or {}
                chunks.append({"id": self.memory.id_of(r), "time": meta.get("time"), "content": meta.get("content"),
                               "score": float(s), "source": "vector"})
            return chunks
        except _lib.VidmemError as e:
            if e.code == _lib.VM_ERR_INVALID:
                raise
            logger.warning("Vector search failed: %s", e)
            return []
        except Exception as e:  # embedder failure etc.: reference returns [] (retriever_hybrid.py:321-323)
            logger.warning("Vector search failed: %s", e)
            return []

    async def _post_compress_chunks(self, query, chunks: List[Dict]) -> List[Dict]:
        if not self.embedder or not chunks:
            return chunks
        try:
            query_embedding = await self.embedder.aembed_query(query)
            segments, owners = [], []
            for chunk in chunks:
                for segment in (self.splitter(chunk["content"]) if self.splitter else [chunk["content"]]):
                    segments.append(segment)
                    owners.append(chunk)
            if not segments:
                return []
            # segment embeds: one failure drops that segment only (retriever_hybrid.py:505-507)
            seg_emb, keep = [], []
            for i, seg in enumerate(segments):
                try:
                    seg_emb.append(await self.embedder.aembed_query(seg))
                    keep.append(i)
                except Exception as e:
                    logger.debug("Failed to embed segment: %s", e)
            if not seg_emb:
                return []
            # ONE [1, D] x [S, D] exact-cosine launch for all segments of all hits (the reference scores them one
            # by one in Python, :497); filter >= threshold in encounter order, then [:top_k] (:499-510).  The operands
            # are scored as fp32 values, not rounded to the memory's 16-bit type: they never enter the memory, and a
            # threshold decision must not flip on a rounding the reference does not make.  A segment whose embedding's
            # length differs from the query's (an embedder that changed its model between calls) is scored on the host
            # by the reference's own zip-truncating expression (:655-664: dot over the common prefix, each magnitude over
            # its whole vector, ``mag1 * mag2 == 0`` -> 0.0) - three Python sums, as there.
            same = [j for j, e in enumerate(seg_emb) if len(e) == len(query_embedding)]
            sims = [0.0] * len(seg_emb)
            if same:
                dev = self.memory.cosine_exact([query_embedding], [seg_emb[j] for j in same], as_f32=True)[0].cpu().tolist()
                for j, v in zip(same, dev):
                    sims[j] = v
            for j, e in enumerate(seg_emb):
                if len(e) != len(query_embedding):
                    sims[j] = _zip_truncating_cosine(query_embedding, e)
            kept = [{**owners[i], "content": segments[i], "compression_score": float(sim)}
                    for i, sim in zip(keep, sims) if sim >= self.config.compression_threshold]
            return kept[: self.config.top_k]
        except Exception as e:
            logger.warning("Post-compression failed: %s", e)
            return chunks
