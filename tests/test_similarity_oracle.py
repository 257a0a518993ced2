"""CPU: the similarity oracle against the golden vectors captured from the reference
(tests/golden/make_similarity_golden.py; reference src/utils/embedding_utils.py:29-39 imported there)."""
import os
import zlib

import numpy as np
import pytest

from oracle import cref
from oracle import similarity_ref as S
from tests.golden.make_similarity_golden import CASES, make_inputs


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "similarity_golden.npz"))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_numpy_and_c_oracle_match_reference_golden(case, golden):
    q, m = make_inputs(case)
    n = case["name"]
    # the seeded inputs are the ones the goldens were captured on
    assert [zlib.crc32(q.tobytes()), zlib.crc32(m.tobytes())] == golden[n + "/input_crc"].tolist()
    rows, scores = S.batch_topk_np(q, m, case["k"])
    assert np.array_equal(rows, golden[n + "/rows"])
    assert np.array_equal(scores, golden[n + "/scores"])  # bit for bit
    c_rows, c_scores = cref.cosine_topk(q.astype(np.float32), m.astype(np.float32), case["k"], dtype="f32")
    assert np.array_equal(c_rows, golden[n + "/rows"])
    assert np.array_equal(c_scores, golden[n + "/scores"])
    # the reference's own **0.5 flavour agrees to the last ulp
    assert np.allclose(scores, golden[n + "/scores_ref_utils"], rtol=4e-16, atol=0)


def test_pure_python_restatement_small_case(golden):
    case = CASES[-1]  # 3 x 7, k=10 > M: short lists, -1 padding
    q, m = make_inputs(case)
    ids = {f"row{r}": m[r].tolist() for r in range(case["M"])}
    out = S.calculate_batch_similarities_ref([r.tolist() for r in q], ids, case["k"])
    rows = golden[case["name"] + "/rows"]
    scores = golden[case["name"] + "/scores"]
    for qi, lst in enumerate(out):
        assert len(lst) == case["M"]
        assert [int(c[3:]) for c, _ in lst] == rows[qi, : len(lst)].tolist()
        assert [s for _, s in lst] == scores[qi, : len(lst)].tolist()
        assert (rows[qi, len(lst):] == -1).all()
    merged = S.merge_batch_similarities_ref(out, 2)
    assert [int(c[3:]) for c, _ in merged] == golden[case["name"] + "/merged_rows"].tolist()
    assert [s for _, s in merged] == golden[case["name"] + "/merged_scores"].tolist()


def test_exception_entries_and_guards():
    # src/components/pre_llm_injector.py:357-359 (Exception -> []), :378-379 (len mismatch), :385-386 (zero norm)
    mem = {"a": [1.0, 0.0], "b": [0.0, 0.0], "c": [], "d": [1.0, 1.0]}
    out = S.calculate_batch_similarities_ref([RuntimeError("embed failed"), [1.0, 0.0]], mem, 3)
    assert out[0] == []
    assert [c for c, _ in out[1]] == ["a", "d", "b"]  # "c" (empty embedding) is skipped, zero row scores 0.0
    assert out[1][2][1] == 0.0
    assert S.cosine_similarity_ref([1.0, 2.0], [1.0]) == 0.0
    assert S.cosine_similarity_ref([1.0, 2.0], [1.0], variant="hybrid") == 1.0 / (5.0 ** 0.5)  # zip-truncation


def test_vector_search_threshold_and_modes(golden):
    case = CASES[0]
    q, m = make_inputs(case)
    vs = S.vector_search_ref(q[0].tolist(), [r.tolist() for r in m], limit=case["k"], min_score=0.3)
    assert [r for r, _ in vs] == golden[case["name"] + "/vs_rows"].tolist()
    assert all(s > 0.3 for _, s in vs)
    rows, scores = S.batch_topk_np(q[:1], m, case["k"], min_score=0.3)
    assert rows[0, : len(vs)].tolist() == [r for r, _ in vs] and (rows[0, len(vs):] == -1).all()
    unit = S.vector_search_ref(q[0].tolist(), [r.tolist() for r in m[:50]], limit=5, min_score=0.3,
                               score_mode=S.SCORE_UNIT_INTERVAL)
    assert len(unit) == 5 and all(0.3 < s <= 1.0 for _, s in unit)  # (1+cos)/2 of ~0 cosines is ~0.5


def test_post_compress_filter_order():
    q = [1.0, 0.0]
    segs = [[1.0, 0.1], [0.0, 1.0], [1.0, 0.0], [0.9, 0.1], [1.0, 0.05]]
    kept = S.post_compress_ref(q, segs, threshold=0.9, top_k=3)
    assert [i for i, _ in kept] == [0, 2, 3]  # encounter order, not ranked; truncated to top_k


def test_c_oracle_16bit_paths():
    rng = np.random.default_rng(0)
    m = rng.standard_normal((300, 256)).astype(np.float16)
    q = rng.standard_normal((5, 256)).astype(np.float16)
    rows, scores = cref.cosine_topk(q, m, 7, dtype="f16")
    r2, s2 = S.batch_topk_np(q.astype(np.float64), m.astype(np.float64), 7)
    assert np.array_equal(rows, r2) and np.array_equal(scores, s2)
