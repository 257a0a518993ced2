"""Pins oracle/similarity_ref.py against the reference and writes tests/golden/similarity_golden.npz.

Run in the authoring container (the reference cannot travel to the GPU box; only this script's OUTPUT does):

    python tests/golden/make_similarity_golden.py

What is imported from the reference: ``src.utils.embedding_utils.EmbeddingUtils.cosine_similarity``
(src/utils/embedding_utils.py:29-39) - the only importable piece of the path here (the call-site classes need
neo4j / langchain, which are absent).  The top-k / merge semantics around it are the oracle's restatement of
src/components/pre_llm_injector.py:346-372 and :238-249, run with the inner cosine DELEGATED to the imported
reference function, and compared against the all-oracle run.

Inputs are regenerated from the seed by the tests (vidmem.synthetic); the file stores expected rows / scores and
a checksum of the quantised input bits.
"""
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vidmem  # noqa: E402,F401
from vidmem import synthetic as syn  # noqa: E402
from oracle import similarity_ref as S  # noqa: E402
from oracle import cref  # noqa: E402

REF = "/root/reference"


def import_reference_cosine():
    cwd = os.getcwd()
    os.makedirs("/tmp/oracle_probe", exist_ok=True)
    os.chdir("/tmp/oracle_probe")  # the reference's get_logger creates ./logs in cwd (src/core/logger.py:42-46)
    sys.path.insert(0, REF)
    try:
        from src.utils.embedding_utils import EmbeddingUtils
    finally:
        sys.path.remove(REF)
        os.chdir(cwd)
    return EmbeddingUtils.cosine_similarity


def quantise(x: np.ndarray, dtype: str) -> np.ndarray:
    """fp32 -> fp16/bf16 -> fp64 (exact widening): the values the GPU memory holds."""
    t = torch.from_numpy(np.ascontiguousarray(x))
    t = t.to(torch.float16 if dtype == "f16" else torch.bfloat16)
    return t.to(torch.float64).numpy()


def make_inputs(case):
    """Shared with tests/test_similarity_oracle.py and the GPU parity tests (imported from here)."""
    D, Q, M, dtype, seed = case["D"], case["Q"], case["M"], case["dtype"], case["seed"]
    mem = syn.unit_rows(seed, "memory", M, D) * np.float32(case.get("mem_scale", 1.0))
    qry = syn.unit_rows(seed, "queries", Q, D)
    if case.get("near", 0):  # make queries close to some memory rows so scores are not all ~0
        for i in range(Q):
            qry[i] = 0.6 * mem[(i * 37) % M] + 0.4 * qry[i]
    for (dst, src) in case.get("dups", []):  # exact duplicate rows -> exact score ties
        mem[dst] = mem[src]
    for r in case.get("zero_rows", []):
        mem[r] = 0.0
    for qi in case.get("zero_queries", []):
        qry[qi] = 0.0
    return quantise(qry, dtype), quantise(mem, dtype)


CASES = [
    dict(name="c1_q16_m1000_k5_d768_f16", D=768, Q=16, M=1000, k=5, dtype="f16", seed=7, near=1,
         dups=[(10, 37), (500, 37), (999, 74)], zero_rows=[3], zero_queries=[]),
    dict(name="ref_q4_m322_k3_d768_f16", D=768, Q=4, M=322, k=3, dtype="f16", seed=8, near=1,
         dups=[(1, 0), (2, 0), (3, 0)], zero_rows=[], zero_queries=[2]),
    dict(name="q16_m4096_k10_d768_f16", D=768, Q=16, M=4096, k=10, dtype="f16", seed=9, near=1,
         dups=[(4000, 37), (4001, 37)], zero_rows=[100], zero_queries=[]),
    dict(name="q1_m4096_k20_d1024_bf16", D=1024, Q=1, M=4096, k=20, dtype="bf16", seed=10, near=1,
         dups=[(5, 0), (6, 0), (7, 0), (8, 0)], zero_rows=[9], zero_queries=[]),
    dict(name="q16_m1000_k5_d1024_bf16_unnormalised", D=1024, Q=16, M=1000, k=5, dtype="bf16", seed=11, near=1,
         mem_scale=3.0, dups=[], zero_rows=[0], zero_queries=[15]),
    dict(name="q3_m7_k10_d768_f16_short_memory", D=768, Q=3, M=7, k=10, dtype="f16", seed=12, near=0,
         dups=[(6, 1)], zero_rows=[], zero_queries=[]),
]


def main():
    ref_cos = import_reference_cosine()
    assert ref_cos([1, 2, 3], [3, 2, 1]) == 0.7142857142857143 and ref_cos([0, 0, 0], [1, 2, 3]) == 0.0
    out = {}
    for case in CASES:
        q, m = make_inputs(case)
        Q, M, k = case["Q"], case["M"], case["k"]
        ql, ml = [r.tolist() for r in q], [r.tolist() for r in m]
        # --- reference-driven run: restated loop, imported cosine ------------------------------------------
        ref_rows = np.full((Q, k), -1, np.int64)
        ref_scores = np.zeros((Q, k), np.float64)
        ref_matrix = np.zeros((Q, M), np.float64)
        for qi in range(Q):
            sims = [(r, float(ref_cos(ql[qi], ml[r]))) for r in range(M)]
            ref_matrix[qi] = [s for _, s in sims]
            sims.sort(key=lambda x: x[1], reverse=True)
            for j, (r, s) in enumerate(sims[:k]):
                ref_rows[qi, j], ref_scores[qi, j] = r, s
        # --- all-oracle runs ----------------------------------------------------------------------------
        ids = {f"row{r}": ml[r] for r in range(M)}
        py = S.calculate_batch_similarities_ref(ql, ids, k)  # math.sqrt variant (the real call site)
        py_rows = np.full((Q, k), -1, np.int64)
        py_scores = np.zeros((Q, k), np.float64)
        for qi, lst in enumerate(py):
            for j, (cid, s) in enumerate(lst):
                py_rows[qi, j], py_scores[qi, j] = int(cid[3:]), s
        np_rows, np_scores = S.batch_topk_np(q, m, k)
        c_rows, c_scores = cref.cosine_topk(q.astype(np.float32), m.astype(np.float32), k, dtype="f32")
        utils_matrix = np.array([[S.cosine_similarity_ref(ql[qi], ml[r], "utils") for r in range(M)]
                                 for qi in range(Q)])
        # pins: (1) the oracle's "utils" variant IS the imported reference, bit for bit
        assert np.array_equal(utils_matrix, ref_matrix), case["name"]
        # (2) math.sqrt vs **0.5 may differ in the last ulp; rows must not
        assert np.array_equal(py_rows, ref_rows), case["name"]
        ulp_diff = int((py_scores != ref_scores).sum())
        assert np.allclose(py_scores, ref_scores, rtol=4e-16, atol=0), case["name"]
        # (3) numpy and C restatements == pure-Python restatement, bit for bit
        assert np.array_equal(np_rows, py_rows) and np.array_equal(np_scores, py_scores), case["name"]
        assert np.array_equal(c_rows, py_rows) and np.array_equal(c_scores, py_scores), case["name"]
        merged = S.merge_batch_similarities_ref(py, 2)
        vs = S.vector_search_ref(ql[0], ml, limit=k, min_score=0.3)
        n = case["name"]
        out[n + "/rows"] = py_rows
        out[n + "/scores"] = py_scores
        out[n + "/scores_ref_utils"] = ref_scores
        out[n + "/merged_rows"] = np.array([int(c[3:]) for c, _ in merged], np.int64)
        out[n + "/merged_scores"] = np.array([s for _, s in merged], np.float64)
        out[n + "/vs_rows"] = np.array([r for r, _ in vs], np.int64)
        out[n + "/vs_scores"] = np.array([s for _, s in vs], np.float64)
        out[n + "/input_crc"] = np.array([zlib.crc32(q.tobytes()), zlib.crc32(m.tobytes())], np.int64)
        print(f"{n}: pinned ({ulp_diff} scores differ in the last ulp between sqrt and **0.5); "
              f"top score {py_scores[0, 0]:.6f}, merged {merged}")
    path = os.path.join(ROOT, "tests", "golden", "similarity_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
