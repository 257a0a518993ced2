"""Hybrid-fusion host logic against the oracle restatement of
src/pipeline/retriever_hybrid.py:173-190, :241-262, and the mixin's call surface."""
import asyncio
import random
from types import SimpleNamespace

import pytest

from oracle import similarity_ref as S
from vidmem import fusion


def _chunks(ids):
    return [{"id": i, "content": f"c{n}", "score": 1.0 - 0.01 * n} for n, i in enumerate(ids)]


def test_rankings_edge_cases():
    ids = ["u_0_3", "u_1_7", "u_2_3", "free-form", None, "u_3_x", "7", "u_4_-2"]
    got = fusion.true_chunk_rankings(_chunks(ids), [3, 7, 9, -2])
    assert got == {3: 1, 7: 2, 9: None, -2: 8}                    # first occurrence wins; "7" alone parses too
    assert got == S.true_chunk_rankings_ref(_chunks(ids), [3, 7, 9, -2])
    assert fusion.parse_chunk_index("a_b_12") == 12 and fusion.parse_chunk_index("a_b_c") is None
    assert fusion.true_chunk_rankings([], [1]) == {1: None}


def test_rankings_and_fusion_randomised_against_oracle():
    rng = random.Random(4)
    for _ in range(200):
        ids = [f"run_{rng.randrange(4)}_{rng.randrange(12)}" if rng.random() < 0.9 else rng.choice(["x", None, "a_b"])
               for _ in range(rng.randrange(0, 15))]
        true = [rng.randrange(14) for _ in range(rng.randrange(0, 5))]
        vec, trav = _chunks(ids[: len(ids) // 2]), _chunks(ids[len(ids) // 2:])
        k = rng.randrange(1, 12)
        fused = fusion.fuse_result_chunks(vec or None, trav or None, k)
        assert fused == S.fuse_result_chunks_ref(vec or None, trav or None, k)
        assert fusion.true_chunk_rankings(fused, true) == S.true_chunk_rankings_ref(fused, true)


def test_mixin_requires_memory_and_routes_calls():
    class Base:                                   # stands in for HybridRetriever's attributes
        def __init__(self):
            self.config = SimpleNamespace(top_k_chunks=3, compression_threshold=0.5, top_k=2)
            self.embedder = object()

    class R(fusion.HipHybridMixin, Base):
        pass

    r = R()
    with pytest.raises(RuntimeError):
        asyncio.run(r._vector_search_chunks(None, "q"))

    class FakeHip:
        async def _vector_search_chunks(self, session, query): return [("vec", query)]
        async def _post_compress_chunks(self, query, chunks): return [("pc", query, len(chunks))]

    r._hip = FakeHip()
    assert asyncio.run(r._vector_search_chunks(None, "q")) == [("vec", "q")]
    assert asyncio.run(r._post_compress_chunks("q", [1, 2])) == [("pc", "q", 2)]
