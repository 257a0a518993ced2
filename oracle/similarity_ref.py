"""CPU ORACLE (test infrastructure, NOT product code) for the similarity / top-k half of the hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product path (the ``vidmem`` package over ``libvidmem.so``) never does.

Every function is a plain restatement of reference arithmetic, cited by file:line relative to the
reference repository root:

* ``cosine_similarity_ref``              <- src/components/pre_llm_injector.py:374-388
                                            (same expression as src/pipeline/retriever_hybrid.py:655-664 and
                                            src/utils/embedding_utils.py:29-39; the guards differ, see ``variant``)
* ``calculate_batch_similarities_ref``   <- src/components/pre_llm_injector.py:346-372
* ``merge_batch_similarities_ref``       <- src/components/pre_llm_injector.py:238-249
* ``vector_search_ref``                  <- src/pipeline/retriever_hybrid.py:293-306 (Cypher semantics: threshold,
                                            ORDER BY score DESC, LIMIT).  The arithmetic of
                                            ``vector.similarity.cosine`` lives inside Neo4j (third party, absent):
                                            PARITY UNPINNED for the score normalisation, so ``score_mode`` is
                                            explicit (RAW cosine or UNIT_INTERVAL=(1+cos)/2) and the tie order
                                            (unspecified by Cypher) is defined as ascending row.
* ``post_compress_ref``                  <- src/pipeline/retriever_hybrid.py:497-510 (filter ``>= threshold`` in
                                            encounter order, then ``[:top_k]``)

Pinning: ``tests/golden/make_similarity_golden.py`` checks ``cosine_similarity_ref`` bit-for-bit against the
reference's own importable ``EmbeddingUtils.cosine_similarity`` (src/utils/embedding_utils.py:29-39) and writes the
golden vectors under ``tests/golden/``.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

SCORE_RAW = 0
SCORE_UNIT_INTERVAL = 1


# --------------------------------------------------------------------------------------------------------------
# pure-Python restatements (slow; small cases only)
# --------------------------------------------------------------------------------------------------------------
def cosine_similarity_ref(vec1: Sequence[float], vec2: Sequence[float], variant: str = "pre_llm") -> float:
    """fp64, three left-to-right generator sums, exactly as the reference.

    variant="pre_llm"  : len mismatch -> 0.0; ``norm1 == 0 or norm2 == 0`` -> 0.0   (pre_llm_injector.py:378-386)
    variant="hybrid"   : zip-truncates; ``mag1 * mag2 == 0`` -> 0.0                  (retriever_hybrid.py:659-663)
    variant="utils"    : zip-truncates; ``** 0.5`` instead of math.sqrt               (embedding_utils.py:31-37)
    """
    if variant == "pre_llm" and len(vec1) != len(vec2):
        return 0.0
    dot_product = sum(a * b for a, b in zip(vec1, vec2))
    if variant == "utils":
        norm1 = sum(a * a for a in vec1) ** 0.5
        norm2 = sum(b * b for b in vec2) ** 0.5
    else:
        norm1 = math.sqrt(sum(a * a for a in vec1))
        norm2 = math.sqrt(sum(b * b for b in vec2))
    if variant == "hybrid":
        if norm1 * norm2 == 0:
            return 0.0
    elif norm1 == 0 or norm2 == 0:
        return 0.0
    return dot_product / (norm1 * norm2)


def calculate_batch_similarities_ref(
    chunk_embeddings: Sequence,
    existing_chunk_embeddings: Dict[str, Sequence[float]],
    top_k: int,
) -> List[List[Tuple[str, float]]]:
    """pre_llm_injector.py:346-372.  ``existing_chunk_embeddings`` is an insertion-ordered dict (memory order)."""
    similarities = []
    for chunk_emb in chunk_embeddings:
        if isinstance(chunk_emb, Exception):
            similarities.append([])
            continue
        chunk_similarities = []
        for chunk_id, existing_emb in existing_chunk_embeddings.items():
            if existing_emb is not None and len(existing_emb) > 0:  # ``if existing_emb:`` for a list
                similarity = cosine_similarity_ref(chunk_emb, existing_emb)
                chunk_similarities.append((chunk_id, float(similarity)))
        chunk_similarities.sort(key=lambda x: x[1], reverse=True)  # stable: ties keep memory order
        similarities.append(chunk_similarities[:top_k])
    return similarities


def merge_batch_similarities_ref(
    batch_similarities: Iterable[Iterable[Tuple[str, float]]], top_k_similar_batch: int
) -> List[Tuple[str, float]]:
    """pre_llm_injector.py:238-249: max score per id (first-seen insertion order), stable desc sort, slice."""
    final_scores: Dict[str, float] = {}
    for chunk_similarities in batch_similarities:
        for chunk_id, score in chunk_similarities:
            if chunk_id not in final_scores or score > final_scores[chunk_id]:
                final_scores[chunk_id] = score
    final_score_list = sorted(final_scores.items(), key=lambda x: x[1], reverse=True)
    return final_score_list[:top_k_similar_batch]


def vector_search_ref(
    query: Sequence[float],
    memory_rows: Sequence[Sequence[float]],
    limit: int,
    min_score: float = 0.3,
    score_mode: int = SCORE_RAW,
) -> List[Tuple[int, float]]:
    """retriever_hybrid.py:293-306: ``WHERE similarity > min_score ... ORDER BY score DESC LIMIT``.

    Returns (row, score).  Tie order = ascending row (build-defined; Cypher leaves it unspecified).
    """
    out = []
    for row, emb in enumerate(memory_rows):
        s = cosine_similarity_ref(query, emb, variant="hybrid")
        if score_mode == SCORE_UNIT_INTERVAL:
            s = (1.0 + s) / 2.0
        if s > min_score:
            out.append((row, s))
    out.sort(key=lambda x: x[1], reverse=True)
    return out[:limit]


def post_compress_ref(
    query: Sequence[float], segments: Sequence[Sequence[float]], threshold: float, top_k: int
) -> List[Tuple[int, float]]:
    """retriever_hybrid.py:497-510: keep segments with cosine >= threshold in encounter order, then [:top_k]."""
    kept = []
    for i, seg in enumerate(segments):
        s = cosine_similarity_ref(query, seg, variant="hybrid")
        if s >= threshold:
            kept.append((i, float(s)))
    return kept[:top_k]


def true_chunk_rankings_ref(chunks, true_chunks):
    """retriever_hybrid.py:173-190 / :247-262: 1-based rank of the first candidate whose id's last ``_`` field parses
    to each true-chunk index; None when none does."""
    rankings = {}
    for idx in true_chunks:
        rankings[idx] = None
    for pos, c in enumerate(chunks, start=1):
        try:
            last_part = str(c.get("id")).split("_")[-1]
            parsed_idx = int(last_part)
        except Exception:
            parsed_idx = None
        if parsed_idx is not None and parsed_idx in rankings and rankings[parsed_idx] is None:
            rankings[parsed_idx] = pos
    return rankings


def fuse_result_chunks_ref(chunk_results, traversal_chunks, top_k_chunks):
    """retriever_hybrid.py:241-244: concatenation, then ``[:top_k_chunks]``."""
    result_chunks = (chunk_results or []) + (traversal_chunks or [])
    return result_chunks[:top_k_chunks]


# --------------------------------------------------------------------------------------------------------------
# numpy restatement with the SAME arithmetic (sequential fp64 sums) for mid-size cases
# --------------------------------------------------------------------------------------------------------------
def _seq_sum_rows(x: np.ndarray) -> np.ndarray:
    """Left-to-right fp64 sum along the last axis, starting from int 0 like Python's ``sum`` (0 + x0 == x0)."""
    # np.add.accumulate is strictly sequential (unlike np.sum, which is pairwise).
    return np.add.accumulate(x, axis=-1, dtype=np.float64)[..., -1]


def cosine_matrix_np(queries: np.ndarray, memory: np.ndarray, row_block: int = 8192) -> np.ndarray:
    """[Q,D] x [M,D] -> [Q,M] fp64 cosine, bit-identical to ``cosine_similarity_ref`` pair by pair."""
    q = np.asarray(queries, dtype=np.float64)
    m = np.asarray(memory, dtype=np.float64)
    Q, D = q.shape
    M = m.shape[0]
    out = np.zeros((Q, M), dtype=np.float64)
    if M == 0 or Q == 0:
        return out
    qn = np.sqrt(_seq_sum_rows(q * q))
    for r0 in range(0, M, row_block):
        mb = m[r0:r0 + row_block]
        mn = np.sqrt(_seq_sum_rows(mb * mb))
        for qi in range(Q):
            dot = _seq_sum_rows(q[qi][None, :] * mb)
            zero = (qn[qi] == 0) | (mn == 0)
            den = qn[qi] * mn
            den[zero] = 1.0
            s = dot / den
            s[zero] = 0.0
            out[qi, r0:r0 + row_block] = s
    return out


def topk_np(scores: np.ndarray, k: int, min_score=None, strict: bool = True):
    """Stable descending top-k per query over a [Q,M] score matrix -> (rows[Q,k] int64 (-1 pad), scores[Q,k] fp64).

    Ties keep ascending row order (Python's stable ``sort(reverse=True)`` on memory-ordered input).
    ``min_score`` filters ``> min_score`` (strict, retriever_hybrid.py:298) or ``>=`` (strict=False).
    """
    Q, M = scores.shape
    rows = np.full((Q, k), -1, dtype=np.int64)
    vals = np.zeros((Q, k), dtype=np.float64)
    for qi in range(Q):
        s = scores[qi]
        order = np.argsort(-s, kind="stable")  # stable on negated keys == stable descending
        if min_score is not None:
            keep = s[order] > min_score if strict else s[order] >= min_score
            order = order[keep]
        order = order[:k]
        rows[qi, : order.size] = order
        vals[qi, : order.size] = s[order]
    return rows, vals


def batch_topk_np(queries, memory, k, min_score=None, score_mode=SCORE_RAW):
    s = cosine_matrix_np(queries, memory)
    if score_mode == SCORE_UNIT_INTERVAL:
        s = (1.0 + s) / 2.0
    return topk_np(s, k, min_score=min_score)
