"""GPU: the drop-in call surfaces themselves - same names, return shapes and error convention as the
reference's methods - against the pure-Python oracle on the stored 16-bit values:
  _calculate_batch_similarities  src/components/pre_llm_injector.py:346-372 (+ max-merge :238-249)
  _vector_search_chunks          src/pipeline/retriever_hybrid.py:284-323
  _post_compress_chunks          src/pipeline/retriever_hybrid.py:465-514
Bar: ids identical, scores bit-identical fp64."""
import asyncio
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import similarity_ref as S

pytestmark = pytest.mark.gpu

D = 128


def _f16_lists(x):
    return torch.tensor(x, dtype=torch.float32).to(torch.float16).to(torch.float64).tolist()


def _memory(n=60, seed=0, dup=True):
    from vidmem.memory import EmbeddingMemory
    rng = np.random.default_rng(seed)
    rows = rng.standard_normal((n, D))
    if dup:
        rows[17] = rows[3]          # exact tie: memory order must decide
        rows[40] = 0.0              # zero row: cosine guard -> 0.0
    mem = EmbeddingMemory(128, D, "f16")
    ids = [f"run_{i // 8}_{i % 8}" for i in range(n)]
    mem.append(rows.tolist(), ids=ids, meta=[{"time": f"00:{i:02d}-00:{i + 1:02d}", "content": f"text {i}"}
                                             for i in range(n)])
    return mem, ids, _f16_lists(rows)


class DictEmbedder:
    """async aembed_query(text) -> List[float] from a table; raises for unknown texts (a failed remote call)."""

    def __init__(self, table):
        self.table = table

    async def aembed_query(self, text):
        if text not in self.table:
            raise RuntimeError(f"embed failed for {text!r}")
        return self.table[text]


def test_calculate_batch_similarities_matches_reference_loop():
    from vidmem.similarity import HipPreLLMSimilarity, merge_batch_similarities
    mem, ids, rows16 = _memory()
    rng = np.random.default_rng(1)
    q = rng.standard_normal((5, D))
    q[1] = np.asarray(rows16[3]) * 0.5 + q[1] * 0.1
    q16 = _f16_lists(q)
    chunk_embeddings = [q16[0], RuntimeError("embed failed"), q16[1], q16[2], q16[3], q16[4]]
    sim = HipPreLLMSimilarity(mem, SimpleNamespace(top_k_chunk_with_batch_similarity=3))
    got = asyncio.run(sim._calculate_batch_similarities(chunk_embeddings, neo4j_handler=None))
    want = S.calculate_batch_similarities_ref(chunk_embeddings, dict(zip(ids, rows16)), 3)
    assert got == want and got[1] == []                           # ids, fp64 scores, [] for the failed embed
    assert merge_batch_similarities(got, 2) == S.merge_batch_similarities_ref(want, 2)
    # a query of the wrong length scores 0.0 against every row in the reference (pre_llm_injector.py:378-379); the
    # stable sort then keeps memory order: first k chunks, score 0.0 - not an exception
    short = q16[0][:D - 5]
    mixed = [q16[0], short, q16[1]]
    got_m = asyncio.run(sim._calculate_batch_similarities(mixed, neo4j_handler=None))
    want_m = S.calculate_batch_similarities_ref(mixed, dict(zip(ids, rows16)), 3)
    assert got_m == want_m and got_m[1] == [(ids[0], 0.0), (ids[1], 0.0), (ids[2], 0.0)]
    # an empty memory gives empty lists, not an error (first batch of a run)
    from vidmem.memory import EmbeddingMemory
    empty = HipPreLLMSimilarity(EmbeddingMemory(16, D, "f16"), SimpleNamespace(top_k_chunk_with_batch_similarity=3))
    assert asyncio.run(empty._calculate_batch_similarities([q16[0]], None)) == [[]]


@pytest.mark.parametrize("mode", ["raw", "unit"])
def test_vector_search_chunks_matches_reference_query(mode):
    from vidmem import _lib
    from vidmem.similarity import HipVectorSearch
    mem, ids, rows16 = _memory()
    rng = np.random.default_rng(2)
    q16 = _f16_lists((np.asarray(rows16[3]) + 0.8 * rng.standard_normal(D))[None])[0]
    sm = _lib.VM_SCORE_RAW if mode == "raw" else _lib.VM_SCORE_UNIT_INTERVAL
    thr = 0.3 if mode == "raw" else 0.55
    vs = HipVectorSearch(mem, DictEmbedder({"who?": q16}), SimpleNamespace(top_k_chunks=6), min_score=thr,
                         score_mode=sm)
    got = asyncio.run(vs._vector_search_chunks(None, "who?"))
    want = S.vector_search_ref(q16, rows16, 6, min_score=thr,
                               score_mode=S.SCORE_RAW if mode == "raw" else S.SCORE_UNIT_INTERVAL)
    assert [(c["id"], c["score"]) for c in got] == [(ids[r], s) for r, s in want] and len(got) >= 2
    assert all(set(c) == {"id", "time", "content", "score", "source"} and c["source"] == "vector" for c in got)
    assert got[0]["content"] == f"text {want[0][0]}" and got[0]["time"].startswith("00:")
    # embedder failure -> [] (retriever_hybrid.py:321-323), never an exception
    assert asyncio.run(vs._vector_search_chunks(None, "unknown text")) == []


def test_post_compress_chunks_matches_reference_filter():
    from vidmem.similarity import HipVectorSearch
    mem, ids, rows16 = _memory(dup=False)
    rng = np.random.default_rng(3)
    q = rng.standard_normal(D)
    segs = {}
    chunks = []
    for c in range(4):
        parts = []
        for s_i in range(3):
            name = f"seg {c}.{s_i}"
            mix = 0.9 if (c + s_i) % 2 == 0 else 0.05
            segs[name] = _f16_lists((mix * q + (1 - mix) * rng.standard_normal(D))[None])[0]
            parts.append(name)
        chunks.append({"id": ids[c], "time": "t", "content": "|".join(parts), "score": 0.9 - 0.1 * c,
                       "source": "vector"})
    del segs["seg 1.1"]                                  # this segment's embed call fails: skipped, not fatal (:505-507)
    # two segments come back with a DIFFERENT length than the query (the embedder changed its model): the reference's
    # cosine zip-truncates (:655-664: dot over the common prefix, each magnitude over its own vector) - one of them ends
    # up above the threshold, one below, and both must be treated exactly as the reference treats them
    segs["seg 0.2"] = segs["seg 0.0"] + [0.001] * 7      # longer: same prefix as a kept segment, slightly larger norm
    segs["seg 2.1"] = segs["seg 2.0"][: D // 2]          # shorter: half of a kept segment
    q16 = _f16_lists(q[None])[0]
    table = dict(segs, **{"the query": q16})
    cfg = SimpleNamespace(top_k_chunks=6, compression_threshold=0.6, top_k=4)
    from vidmem import _lib
    vs = HipVectorSearch(mem, DictEmbedder(table), cfg, score_mode=_lib.VM_SCORE_RAW,
                         splitter=lambda text: text.split("|"))
    got = asyncio.run(vs._post_compress_chunks("the query", chunks))
    order = [(c, name) for c in range(4) for name in chunks[c]["content"].split("|") if name in segs]
    want = S.post_compress_ref(q16, [segs[name] for _, name in order], threshold=0.6, top_k=4)
    want_rows = [(ids[order[i][0]], order[i][1], s) for i, s in want]
    assert [(g["id"], g["content"], g["compression_score"]) for g in got] == want_rows
    assert len(got) == 4 and all(g["source"] == "vector" and "score" in g for g in got)   # {**chunk, ...} keeps keys
    # query embed failure -> the chunks come back unchanged (:512-514); no chunks -> returned as is (:467-468)
    assert asyncio.run(vs._post_compress_chunks("unknown", chunks)) == chunks
    assert asyncio.run(vs._post_compress_chunks("the query", [])) == []


def test_hybrid_mixin_on_a_retriever_shaped_class():
    from vidmem import fusion
    mem, ids, rows16 = _memory()
    q16 = rows16[10]

    class Base:
        def __init__(self):
            self.config = SimpleNamespace(top_k_chunks=4, compression_threshold=0.5, top_k=3)
            self.embedder = DictEmbedder({"q": q16})

    class GpuRetriever(fusion.HipHybridMixin, Base):
        pass

    from vidmem import _lib
    with pytest.raises(TypeError):
        GpuRetriever().attach_memory(mem)                   # the score mapping is never guessed
    r = GpuRetriever().attach_memory(mem, score_mode=_lib.VM_SCORE_RAW, min_score=0.3)
    hits = asyncio.run(r._vector_search_chunks(None, "q"))
    assert hits[0]["id"] == ids[10] and abs(hits[0]["score"] - 1.0) < 1e-12
    fused = fusion.fuse_result_chunks(hits, [{"id": "run_9_5"}], r.config.top_k_chunks)
    ranks = fusion.true_chunk_rankings(fused, [int(ids[10].split("_")[-1]), 5])
    assert ranks[int(ids[10].split("_")[-1])] == 1


def test_post_compress_scores_unrounded_fp32_embeddings():
    """The post-compression operands are fresh embedder outputs (fp32 values from an OpenAI-compatible server) that
    never enter the 16-bit memory: they are scored as they are (VM_F32 path of vm_cosine_exact), bit-identical to the
    reference's Python loop on the same floats (src/pipeline/retriever_hybrid.py:497), and a segment whose cosine sits
    between the threshold and its fp16-rounded value is decided as the reference decides it."""
    from vidmem import _lib
    from vidmem.similarity import HipVectorSearch
    mem, ids, _ = _memory(dup=False)
    rng = np.random.default_rng(17)
    Dq = 96                                                    # need not equal the memory's dimension
    q = rng.standard_normal(Dq).astype(np.float32)
    segs = {f"s{i}": (0.6 * q + 0.8 * rng.standard_normal(Dq).astype(np.float32)).astype(np.float32) for i in range(12)}
    f = lambda a: [float(v) for v in a]                        # fp32 values as Python floats, as JSON delivers them
    exact = {n: S.cosine_similarity_ref(f(q), f(v), variant="hybrid") for n, v in segs.items()}
    r16 = lambda a: f(np.asarray(a, np.float32).astype(np.float16).astype(np.float32))
    rounded = {n: S.cosine_similarity_ref(r16(q), r16(v), variant="hybrid") for n, v in segs.items()}
    # a threshold strictly between the exact and the fp16-rounded score of one segment: the two scorings disagree there
    name = max(segs, key=lambda n: abs(exact[n] - rounded[n]))
    thr = 0.5 * (exact[name] + rounded[name])
    assert (exact[name] >= thr) != (rounded[name] >= thr)
    chunks = [{"id": ids[0], "time": "t", "content": "|".join(segs), "score": 0.9, "source": "vector"}]
    table = {n: f(v) for n, v in segs.items()}
    table["the query"] = f(q)
    cfg = SimpleNamespace(top_k_chunks=6, compression_threshold=thr, top_k=50)
    vs = HipVectorSearch(mem, DictEmbedder(table), cfg, score_mode=_lib.VM_SCORE_RAW,
                         splitter=lambda text: text.split("|"))
    got = asyncio.run(vs._post_compress_chunks("the query", chunks))
    order = list(segs)
    want = S.post_compress_ref(f(q), [f(segs[n]) for n in order], threshold=thr, top_k=50)
    assert [(g["content"], g["compression_score"]) for g in got] == [(order[i], s) for i, s in want]
    assert (name in [g["content"] for g in got]) == (exact[name] >= thr)
