"""GPU parity: libvidmem's cosine top-k (through the C ABI) against the oracle and the reference-derived goldens.

Bar: rows bit-identical, scores bit-identical fp64 (the kernel re-scores its candidates with the reference's own
sequence of roundings).  Sizes: goldens (<= 4096 rows), a 100k-row seeded case against the C oracle, and 1M-row
size-independent properties.
"""
import os

import numpy as np
import pytest
import torch

from oracle import cref
from oracle import similarity_ref as S
from tests.golden.make_similarity_golden import CASES, make_inputs

pytestmark = pytest.mark.gpu

TD = {"f16": torch.float16, "bf16": torch.bfloat16}


def _mem(case_or_dtype, capacity, dim, ring=False):
    from vidmem.memory import EmbeddingMemory
    return EmbeddingMemory(capacity, dim, dtype=case_or_dtype, ring=ring)


def _bits(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "similarity_golden.npz"))


@pytest.mark.parametrize("exact", [False, True], ids=["scan", "exhaustive"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_golden_cases_bit_exact(case, exact, golden):
    q, m = make_inputs(case)
    mem = _mem(case["dtype"], max(case["M"], 16), case["D"])
    mem.append(torch.tensor(m, dtype=torch.float64))
    scores, rows = mem.topk(torch.tensor(q, dtype=torch.float64), case["k"], exact=exact)
    n = case["name"]
    assert np.array_equal(rows.cpu().numpy(), golden[n + "/rows"])
    assert np.array_equal(scores.cpu().numpy(), golden[n + "/scores"])
    if not exact:
        assert mem.uncertified_count >= 0


def test_all_pairs_exact_matches_c_oracle():
    rng = np.random.default_rng(5)
    for dtype, D in (("f16", 768), ("bf16", 1024)):
        m = torch.tensor(rng.standard_normal((1000, D)) * 0.7, dtype=torch.float32).to(TD[dtype])
        q = torch.tensor(rng.standard_normal((9, D)), dtype=torch.float32).to(TD[dtype])
        m[17] = 0
        q[3] = 0
        mem = _mem(dtype, 16, D)
        got = mem.cosine_exact(q.cuda(), m.cuda()).cpu().numpy()
        want = cref.cosine_matrix(_bits(q), _bits(m), dtype=dtype)
        assert np.array_equal(got, want)  # bit for bit, including the zero-norm guards


def test_threshold_and_score_mode():
    case = CASES[0]
    q, m = make_inputs(case)
    mem = _mem("f16", case["M"], case["D"])
    mem.append(torch.tensor(m, dtype=torch.float64))
    qt = torch.tensor(q, dtype=torch.float64)
    for mode in (S.SCORE_RAW, S.SCORE_UNIT_INTERVAL):
        for thr in (0.3, 0.52, 0.9):
            want_r, want_s = S.batch_topk_np(q, m, 8, min_score=thr, score_mode=mode)
            for exact in (False, True):
                s, r = mem.topk(qt, 8, min_score=thr, score_mode=mode, exact=exact)
                assert np.array_equal(r.cpu().numpy(), want_r), (mode, thr, exact)
                assert np.array_equal(s.cpu().numpy(), want_s), (mode, thr, exact)


@pytest.mark.parametrize("M,Q,k", [(0, 3, 5), (1, 1, 1), (5, 2, 8), (17, 17, 3), (33, 65, 12), (300, 130, 26),
                                   (200, 4, 40)])
def test_ragged_shapes(M, Q, k):
    rng = np.random.default_rng(M * 1000 + Q)
    D = 256
    m = torch.tensor(rng.standard_normal((max(M, 1), D)), dtype=torch.float32).to(torch.float16)[:M]
    q = torch.tensor(rng.standard_normal((Q, D)), dtype=torch.float32).to(torch.float16)
    mem = _mem("f16", max(M, 16), D)
    if M:
        mem.append(m)
    s, r = mem.topk(q, k)
    want_r, want_s = cref.cosine_topk(_bits(q), _bits(m) if M else np.zeros((0, D), np.uint16), k, dtype="f16")
    assert np.array_equal(r.cpu().numpy(), want_r)
    assert np.array_equal(s.cpu().numpy(), want_s)


def test_ring_memory_overwrites_oldest():
    rng = np.random.default_rng(11)
    D, cap = 256, 96
    mem = _mem("f16", cap, D, ring=True)
    allrows = torch.tensor(rng.standard_normal((250, D)), dtype=torch.float32).to(torch.float16)
    q = torch.tensor(rng.standard_normal((5, D)), dtype=torch.float32).to(torch.float16)
    done = 0
    for step in (40, 56, 30, 64, 60):  # crosses the wrap point several times
        first = mem.append(allrows[done:done + step])
        assert first == done
        done += step
        lo = max(0, done - cap)
        want_r, want_s = cref.cosine_topk(_bits(q), _bits(allrows[lo:done]), 6, dtype="f16")
        want_r = np.where(want_r >= 0, want_r + lo, -1)
        for exact in (False, True):
            s, r = mem.topk(q, 6, exact=exact)
            assert np.array_equal(r.cpu().numpy(), want_r), (done, exact)
            assert np.array_equal(s.cpu().numpy(), want_s), (done, exact)


def test_many_exact_ties_fall_back_to_exhaustive():
    # 40 identical rows: more ties than candidate slots -> the scan must refuse to certify, the wrapper must
    # still return the reference answer (lowest row ids first).
    rng = np.random.default_rng(3)
    D = 256
    base = torch.tensor(rng.standard_normal((1, D)), dtype=torch.float32).to(torch.float16)
    m = torch.tensor(rng.standard_normal((500, D)), dtype=torch.float32).to(torch.float16)
    m[100:140] = base
    mem = _mem("f16", 512, D)
    mem.append(m)
    s, r = mem.topk(base, 10)
    assert r.cpu().numpy().tolist() == [list(range(100, 110))]
    assert mem.uncertified_count == 1
    want_r, want_s = cref.cosine_topk(_bits(base), _bits(m), 10, dtype="f16")
    assert np.array_equal(s.cpu().numpy(), want_s)
    # the scan alone (redo=False) is NOT the reference answer here: that is what the flags are for
    s0, r0 = mem.topk(base, 10, redo=False)
    assert set(r0[0].tolist()) <= set(range(100, 140))


def test_the_shape_that_must_redo_near_ties_around_rank_k():
    """The one configuration the fast path REFUSES to certify without its candidate buffers overflowing, pinned down in
    round 4 (tools/uncert_probe.py: a clip processed twice - every stored row twice - made the 10th and the 17th best
    score of a per-chunk search 9.2e-5 apart): the exact k-th score must clear the best score a KL-entry list can have
    REJECTED by the fp32 error bound 2 (D + 8) 2^-24; with KL - k + 1 or more rows within that bound of rank k it
    cannot, whatever the scan variant.  Such a query is flagged VM_FLAG_GAP (1, not the overflow code 3), redone
    exhaustively on the device, and the answer is the reference's.  Fewer near-ties than the slack: certified."""
    from vidmem import _lib
    rng = np.random.default_rng(21)
    D, k = 256, 10                                            # k = 10 -> lists of KL = 16: six ranks of slack
    base = torch.tensor(rng.standard_normal(D), dtype=torch.float32).to(torch.float16)

    def variants(n):      # rows one fp16 ulp away from `base` in one element each: cosines within ~1e-7 of 1, all different
        v = base.repeat(n, 1).clone()
        bits = v.view(torch.int16)
        for j in range(n):
            bits[j, 3 * j + 1] += 1
        return v
    for near, must_redo in ((20, True), (5, False)):
        m = torch.tensor(rng.standard_normal((500, D)), dtype=torch.float32).to(torch.float16)
        m[200] = base
        m[300:300 + near] = variants(near)
        mem = _mem("f16", 512, D)
        mem.append(m)
        q = torch.stack([base, m[7]])                         # query 1 is an ordinary one
        s, r = mem.topk(q, k)
        flags = mem._scratch.flags[:2].cpu().tolist()
        want_r, want_s = cref.cosine_topk(_bits(q), _bits(m), k, dtype="f16")
        assert np.array_equal(r.cpu().numpy(), want_r) and np.array_equal(s.cpu().numpy(), want_s)
        assert flags[1] == 0
        if must_redo:
            assert flags[0] == _lib.VM_FLAG_GAP and mem.uncertified_count == 1
            gap = float(want_s[0, k - 1] - np.sort(cref.cosine_matrix(_bits(q[:1]), _bits(m), dtype="f16")[0])[::-1][16])
            assert 0.0 <= gap < 2.0 * (D + 8) * 2.0 ** -24       # that is WHY: rank k and rank KL + 1 within the bound
        else:
            assert flags[0] == 0 and mem.uncertified_count == 0


@pytest.mark.parametrize("dtype,D", [("f16", 256), ("bf16", 1024)])
def test_flagged_redo_only_touches_flagged_queries(dtype, D):
    """A batch where SOME queries are uncertifiable (40 exact ties, near-ties below the fp32 bound) and the others
    are ordinary: every query must equal the C oracle bit for bit, the counter must say how many were redone, on a
    ring that has wrapped (device-side row count and head), with min_score / score_mode applied by the redo too."""
    rng = np.random.default_rng(8)
    cap, k = 3000, 12
    hist = torch.tensor(rng.standard_normal((4200, D)), dtype=torch.float32).to(TD[dtype])
    base = hist[3500].clone()
    hist[3600:3640] = base                       # 40 exact duplicates inside the surviving window
    q = torch.tensor(rng.standard_normal((70, D)), dtype=torch.float32).to(TD[dtype])
    q[5] = base
    q[66] = base
    q[30] = 0
    mem = _mem(dtype, cap, D, ring=True)
    for lo in range(0, 4200, 700):
        mem.append(hist[lo:lo + 700])
    live = hist[4200 - cap:]
    for kw in (dict(), dict(min_score=0.05, score_mode=S.SCORE_UNIT_INTERVAL)):
        mem.reset_uncertified()
        s, r = mem.topk(q, k, **kw)
        want_r, want_s = cref.cosine_topk(_bits(q), _bits(live), k, dtype=dtype)
        if kw:
            shown = (1.0 + want_s) / 2.0
            keep = (want_r >= 0) & (shown > 0.05)
            want_s = np.where(keep, shown, 0.0)
            want_r = np.where(keep, want_r, -1)
        want_r = np.where(want_r >= 0, want_r + (4200 - cap), -1)
        assert np.array_equal(r.cpu().numpy(), want_r)
        assert np.array_equal(s.cpu().numpy(), want_s)
        assert mem.uncertified_count >= 2          # at least the two tie queries
    assert r[5].tolist() == [3500] + list(range(3600, 3600 + k - 1))   # stable order: lowest row ids first


def test_100k_rows_against_c_oracle():
    from vidmem import synthetic as syn
    D, M, Q, k = 768, 100_000, 16, 10
    m = torch.from_numpy(syn.unit_rows(7, "memory100k", M, D)).to(torch.float16)
    q = torch.from_numpy(syn.unit_rows(11, "queries100k", Q, D)).to(torch.float16)
    q[:4] = (0.5 * m[[5, 77_777, 99_999, 31_415]].float() + 0.5 * q[:4].float()).to(torch.float16)
    m[60_000] = m[5]  # an exact duplicate far away: tie broken by row id
    mem = _mem("f16", M, D)
    for lo in range(0, M, 25_000):
        mem.append(m[lo:lo + 25_000])
    s, r = mem.topk(q, k)
    assert mem.uncertified_count == 0
    want_r, want_s = cref.cosine_topk(_bits(q), _bits(m), k, dtype="f16")
    assert np.array_equal(r.cpu().numpy(), want_r)
    assert np.array_equal(s.cpu().numpy(), want_s)


def test_full_size_properties_1m_rows():
    """BASELINE metric size (1M x 768): size-independent properties instead of an oracle pass."""
    D, M, Q, k = 768, 1_000_000, 16, 10
    g = torch.Generator(device="cuda").manual_seed(1234)
    m = torch.randn((M, D), generator=g, device="cuda", dtype=torch.float32)
    m = (m / m.norm(dim=1, keepdim=True)).to(torch.float16)
    q = torch.randn((Q, D), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
    planted = [3, 500_000, 999_999, 123_456]
    for i, row in enumerate(planted):
        q[i] = m[row]  # a stored row queried with itself must come first with score == 1 (to fp64 rounding)
    from vidmem.memory import EmbeddingMemory, topk_merge
    mem = EmbeddingMemory(M, D, "f16")
    mem.append(m)
    s, r = mem.topk(q, k)
    assert mem.uncertified_count == 0
    s_np, r_np = s.cpu().numpy(), r.cpu().numpy()
    for i, row in enumerate(planted):
        assert r_np[i, 0] == row and abs(s_np[i, 0] - 1.0) < 1e-12
    assert (np.diff(s_np, axis=1) <= 0).all() and (r_np >= 0).all()
    for i in range(Q):
        assert len(set(r_np[i].tolist())) == k
    # returned scores are the exact reference cosines of the returned rows
    ex = mem.cosine_exact(q[:2], m[r[0]]).cpu().numpy()
    assert np.array_equal(ex[0], s_np[0])
    # sharding invariance: top-k(whole) == merge(top-k(halves)) with global row ids
    halves = []
    for part in range(2):
        hm = EmbeddingMemory(M // 2, D, "f16")
        hm.append(m[part::2])
        halves.append(hm.topk(q, k, row_stride=2, row_offset=part))
        hm.close()
    ms, mr = topk_merge(mem.ctx, torch.stack([h[0] for h in halves]), torch.stack([h[1] for h in halves]))
    assert np.array_equal(mr.cpu().numpy(), r_np) and np.array_equal(ms.cpu().numpy(), s_np)
    # and the exhaustive kernel agrees on one query
    s2, r2 = mem.topk(q[5:6], k, exact=True)
    assert np.array_equal(r2.cpu().numpy(), r_np[5:6]) and np.array_equal(s2.cpu().numpy(), s_np[5:6])


@pytest.mark.parametrize("M", [70_000, 140_000])
def test_sampled_cut_path_many_queries_bit_exact(M):
    """Q >= 32 on a memory >= 65536 rows takes the sampling pre-pass (csrc/topk.hip SAMPLE_ROWS): the cut must not
    change a single row or score bit, including duplicates that straddle the sample boundary."""
    rng = np.random.default_rng(21)
    D, Q, k = 256, 40, 10
    m = torch.tensor(rng.standard_normal((M, D)), dtype=torch.float32).to(torch.float16)
    q = torch.tensor(rng.standard_normal((Q, D)), dtype=torch.float32).to(torch.float16)
    q[:8] = (0.7 * m[[3, 16_383, 16_384, M // 2, M - 1, 5, 6, 7]].float() + 0.3 * q[:8].float()).to(torch.float16)
    m[20_000] = m[3]        # duplicate outside the sample of a row inside it
    m[100] = m[M - 1]       # and the other way round
    q[9] = 0                # zero query: all scores 0.0, first k rows win
    mem = _mem("f16", M, D)
    mem.append(m)
    s, r = mem.topk(q, k)
    want_r, want_s = cref.cosine_topk(_bits(q), _bits(m), k, dtype="f16")
    assert np.array_equal(r.cpu().numpy(), want_r)
    assert np.array_equal(s.cpu().numpy(), want_s)


def test_many_query_groups_bit_exact():
    """Q = 200 at k = 10: four 64-query groups in one 1-D launch (groups of a row block renumbered onto one XCD),
    sampled cut, cross-tile prefetch, network merges - the configuration bench.py's main leg and its Q=256 kNN batch
    run, checked row for row and bit for bit against the C oracle."""
    rng = np.random.default_rng(33)
    D, M, Q, k = 128, 70_000, 200, 10
    m = torch.tensor(rng.standard_normal((M, D)), dtype=torch.float32).to(torch.float16)
    q = torch.tensor(rng.standard_normal((Q, D)), dtype=torch.float32).to(torch.float16)
    q[:50] = (0.6 * m[rng.integers(0, M, 50)].float() + 0.4 * q[:50].float()).to(torch.float16)
    m[65_000] = m[12]          # a duplicate pair far apart
    q[199] = m[12]
    mem = _mem("f16", M, D)
    mem.append(m)
    s, r = mem.topk(q, k)
    assert mem.uncertified_count == 0
    want_r, want_s = cref.cosine_topk(_bits(q), _bits(m), k, dtype="f16")
    assert np.array_equal(r.cpu().numpy(), want_r)
    assert np.array_equal(s.cpu().numpy(), want_s)
    assert r[199, :2].tolist() == [12, 65_000]


@pytest.mark.parametrize("dtype,D,M,Q,k", [("f16", 768, 100_000, 130, 10), ("bf16", 1024, 70_000, 64, 20),
                                            ("f16", 256, 200_000, 256, 10), ("f16", 512, 66_000, 49, 5)])
def test_emit_scan_many_queries_bit_exact(dtype, D, M, Q, k):
    """Q >= 49 on a memory >= 65536 rows takes the query-stationary EMIT scan (csrc/topk_emit.hip): sample cut,
    LDS-staged row tiles, threshold emission, compact, exact re-scoring.  Rows and fp64 scores against the C oracle,
    with planted near-duplicates (top of the list), exact duplicates far apart, a zero query and a zero row, on a
    RING that has wrapped (device-side head / order mapping inside the emit kernel)."""
    rng = np.random.default_rng(D + Q)
    extra = 4_000
    hist = torch.tensor(rng.standard_normal((M + extra, D)), dtype=torch.float32).to(TD[dtype])
    q = torch.tensor(rng.standard_normal((Q, D)), dtype=torch.float32).to(TD[dtype])
    live_lo = extra
    picks = rng.integers(live_lo, M + extra, 20)
    q[:20] = (0.7 * hist[picks].float() + 0.3 * q[:20].float()).to(TD[dtype])
    hist[live_lo + 50_000] = hist[live_lo + 12]         # exact duplicate far apart: lower row id first
    q[21] = hist[live_lo + 12]
    q[22] = 0
    hist[live_lo + 777] = 0
    mem = _mem(dtype, M, D, ring=True)
    for lo in range(0, M + extra, 26_000):
        mem.append(hist[lo:lo + 26_000])
    mem.reset_uncertified()
    s, r = mem.topk(q, k)
    want_r, want_s = cref.cosine_topk(_bits(q), _bits(hist[live_lo:]), k, dtype=dtype)
    want_r = np.where(want_r >= 0, want_r + live_lo, -1)
    assert np.array_equal(r.cpu().numpy(), want_r)
    assert np.array_equal(s.cpu().numpy(), want_s)
    assert r[21, :2].tolist() == [live_lo + 12, live_lo + 50_000]
    assert mem.uncertified_count <= 1                   # the zero query ties with everything: it may be redone


def test_emit_scan_overflow_and_tie_floods_fall_back_to_the_exhaustive_redo():
    """More candidates at or above the cut than the per-query buffer holds (5000 copies of the query > 4096 slots), and
    more exact ties at the KL-th place than the compact kernel ranks (300 copies of another row): the queries must
    be MARKED, redone exhaustively on the device, and still equal the oracle bit for bit; the other queries of the
    batch stay on the fast path."""
    rng = np.random.default_rng(99)
    D, M, Q, k = 256, 70_000, 64, 10
    m = torch.tensor(rng.standard_normal((M, D)), dtype=torch.float32).to(torch.float16)
    q = torch.tensor(rng.standard_normal((Q, D)), dtype=torch.float32).to(torch.float16)
    flood = rng.choice(np.arange(20_000, M), 5000, replace=False)
    m[flood] = q[7].clone()
    ties = rng.choice(np.setdiff1d(np.arange(20_000, M), flood), 300, replace=False)
    m[ties] = m[19_999].clone()
    q[9] = m[19_999]
    mem = _mem("f16", M, D)
    mem.append(m)
    mem.reset_uncertified()
    s, r = mem.topk(q, k)
    want_r, want_s = cref.cosine_topk(_bits(q), _bits(m), k, dtype="f16")
    assert np.array_equal(r.cpu().numpy(), want_r)
    assert np.array_equal(s.cpu().numpy(), want_s)
    assert r[7].tolist() == sorted(flood.tolist())[:k]
    assert 2 <= mem.uncertified_count <= 4


def test_full_size_properties_1m_rows_bf16_k20():
    """BASELINE configs[2] size (1M x 1024 bf16, top-20): size-independent properties instead of an oracle pass -
    planted self-matches first with score 1, strictly ordered distinct rows, the returned scores are the exact
    reference cosines of the returned rows, sharding invariance (top-k(whole) == merge(top-k(halves))), agreement with
    the exhaustive kernel on one query - for Q = 16 (list scan) and Q = 64 (emit scan)."""
    D, M, k = 1024, 1_000_000, 20
    g = torch.Generator(device="cuda").manual_seed(4321)
    from vidmem.memory import EmbeddingMemory, topk_merge
    mem = EmbeddingMemory(M, D, "bf16")
    halves = [EmbeddingMemory(M // 2, D, "bf16") for _ in range(2)]
    keep = []
    for lo in range(0, M, 250_000):
        x = torch.randn((250_000, D), generator=g, device="cuda", dtype=torch.float32)
        x = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
        mem.append(x)
        for part in range(2):
            halves[part].append(x[part::2])        # lo is even: global row r lives in half r % 2 at local row r // 2
        keep.append(x[:8].clone())
    planted = [3, 500_000, 999_999, 250_001]
    rows_of = {3: keep[0][3], 500_000: keep[2][0], 250_001: keep[1][1]}
    rows_of[999_999] = mem.rows_tensor()[999_999].clone()
    for Q in (16, 64):
        q = torch.randn((Q, D), generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
        for i, row in enumerate(planted):
            q[i] = rows_of[row]
        mem.reset_uncertified()
        s, r = mem.topk(q, k)
        s_np, r_np = s.cpu().numpy(), r.cpu().numpy()
        for i, row in enumerate(planted):
            assert r_np[i, 0] == row and abs(s_np[i, 0] - 1.0) < 1e-12, (Q, i)
        assert (np.diff(s_np, axis=1) <= 0).all() and (r_np >= 0).all()
        assert all(len(set(r_np[i].tolist())) == k for i in range(Q))
        ex = mem.cosine_exact(q[5:6], mem.rows_tensor()[r[5]]).cpu().numpy()
        assert np.array_equal(ex[0], s_np[5])
        parts = [h.topk(q, k, row_stride=2, row_offset=p) for p, h in enumerate(halves)]
        ms, mr = topk_merge(mem.ctx, torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
        assert np.array_equal(mr.cpu().numpy(), r_np) and np.array_equal(ms.cpu().numpy(), s_np)
        s2, r2 = mem.topk(q[6:7], k, exact=True)
        assert np.array_equal(r2.cpu().numpy(), r_np[6:7]) and np.array_equal(s2.cpu().numpy(), s_np[6:7])
        assert mem.uncertified_count == 0


def _oracle_rows_parallel(qbits, mbits, k, dtype, picks, threads=8):
    """cref.cosine_topk for a subset of the queries, a few at a time on a thread pool (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    groups = np.array_split(np.asarray(picks), threads)
    with ThreadPoolExecutor(threads) as ex:
        outs = list(ex.map(lambda g: cref.cosine_topk(qbits[g], mbits, k, dtype=dtype), groups))
    return np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])


def test_gemm_class_scan_7040_queries_bit_exact():
    """The query count an 8-GPU step brings to every shard (BASELINE configs[3]: 8 x 880 = 7,040 queries per search)
    takes the GEMM-class scan (csrc/topk_gscan.hip: 256-row x 256-query tiles, emit epilogue, incremental cut cascade)
    on a 100k-row shard.  EVERY query against the exhaustive fp64 kernel bit for bit, and 160 of them - spread over
    every query tile, wave column and lane position, the ragged last tile included - against the C oracle, with near
    duplicates, exact duplicates on both sides of the cascade's pass boundaries, a zero query and a zero row."""
    from vidmem import synthetic as syn
    D, M, Q, k = 768, 100_000, 7040, 10
    m = torch.from_numpy(syn.unit_rows(7, "memory100k", M, D)).to(torch.float16)
    rng = np.random.default_rng(77)
    q = torch.tensor(rng.standard_normal((Q, D)), dtype=torch.float32).to(torch.float16)
    near = rng.integers(0, M, 400)
    q[:400] = (0.6 * m[near].float() + 0.4 * q[:400].float()).to(torch.float16)
    for a, b in ((12, 4095), (12, 4096), (30_000, 32_768), (40_000, 99_999), (5, 60_000)):
        m[b] = m[a]                        # exact duplicates straddling the cascade's limits: lower row id first
    q[401] = m[12]
    q[402] = m[30_000]
    q[7039] = m[40_000]
    q[500] = 0
    m[777] = 0
    mem = _mem("f16", M, D)
    for lo in range(0, M, 25_000):
        mem.append(m[lo:lo + 25_000])
    mem.reset_uncertified()
    s, r = mem.topk(q, k)
    redone = mem.uncertified_count
    s_np, r_np = s.cpu().numpy(), r.cpu().numpy()
    assert r_np[401, :3].tolist() == [12, 4095, 4096] and r_np[402, :2].tolist() == [30_000, 32_768]
    assert r_np[7039, :2].tolist() == [40_000, 99_999]
    s2, r2 = mem.topk(q, k, exact=True)
    assert np.array_equal(r2.cpu().numpy(), r_np) and np.array_equal(s2.cpu().numpy(), s_np)
    picks = np.unique(np.concatenate([np.arange(0, Q, 53), [401, 402, 500, 6911, 6912, 7000, 7039],
                                      256 * np.arange(28) + 17 * (np.arange(28) % 15)]))
    picks = picks[picks < Q]
    want_r, want_s = _oracle_rows_parallel(_bits(q), _bits(m), k, "f16", picks)
    assert np.array_equal(r_np[picks], want_r)
    assert np.array_equal(s_np[picks], want_s)
    assert redone <= 8, redone                 # the zero query ties with everything; nothing else may need the redo


def test_gemm_class_scan_bf16_ring_ragged_bit_exact():
    """GEMM-class scan on a wrapped RING of bf16 rows at D = 1024 with a ragged query count (600 = 2 full tiles + 88)
    and a ragged row count (not a multiple of 256): every query against the exhaustive fp64 kernel, 64 against the C
    oracle."""
    rng = np.random.default_rng(1024)
    D, M, Q, k = 1024, 140_100, 600, 20
    extra = 3_000
    hist = torch.tensor(rng.standard_normal((M + extra, D)), dtype=torch.float32).to(torch.bfloat16)
    q = torch.tensor(rng.standard_normal((Q, D)), dtype=torch.float32).to(torch.bfloat16)
    picks_m = rng.integers(extra, M + extra, 50)
    q[:50] = (0.7 * hist[picks_m].float() + 0.3 * q[:50].float()).to(torch.bfloat16)
    hist[extra + 90_000] = hist[extra + 40]
    q[51] = hist[extra + 40]
    mem = _mem("bf16", M, D, ring=True)
    for lo in range(0, M + extra, 20_000):
        mem.append(hist[lo:lo + 20_000])
    mem.reset_uncertified()
    s, r = mem.topk(q, k)
    s_np, r_np = s.cpu().numpy(), r.cpu().numpy()
    assert r_np[51, :2].tolist() == [extra + 40, extra + 90_000]
    s2, r2 = mem.topk(q, k, exact=True)
    assert np.array_equal(r2.cpu().numpy(), r_np) and np.array_equal(s2.cpu().numpy(), s_np)
    picks = np.unique(np.concatenate([np.arange(0, Q, 11), [51, 511, 512, 599]]))
    want_r, want_s = _oracle_rows_parallel(_bits(q), _bits(hist[extra:]), k, "bf16", picks)
    want_r = np.where(want_r >= 0, want_r + extra, -1)
    assert np.array_equal(r_np[picks], want_r)
    assert np.array_equal(s_np[picks], want_s)
    assert mem.uncertified_count <= 4


def test_gemm_class_scan_tie_flood_marks_and_redoes():
    """5,000 copies of one query's row (more candidates at or above the cut than a candidate buffer holds) and 300
    copies of another row spread over every pass of the incremental cascade: the queries are marked - the mark of an
    EARLY pass must survive the later ones - and redone exhaustively; all 640 queries equal the exhaustive kernel."""
    rng = np.random.default_rng(5)
    D, M, Q, k = 256, 300_000, 640, 10
    m = torch.tensor(rng.standard_normal((M, D)), dtype=torch.float32).to(torch.float16)
    q = torch.tensor(rng.standard_normal((Q, D)), dtype=torch.float32).to(torch.float16)
    flood = rng.choice(np.arange(5_000, 30_000), 5000, replace=False)     # inside the second pass's rows
    m[flood] = q[7].clone()
    ties = rng.choice(np.setdiff1d(np.arange(0, M), flood), 300, replace=False)
    m[ties] = m[M - 1].clone()
    q[9] = m[M - 1]
    mem = _mem("f16", M, D)
    mem.append(m)
    mem.reset_uncertified()
    s, r = mem.topk(q, k)
    assert 2 <= mem.uncertified_count <= 6
    s2, r2 = mem.topk(q, k, exact=True)
    assert np.array_equal(r2.cpu().numpy(), r.cpu().numpy()) and np.array_equal(s2.cpu().numpy(), s.cpu().numpy())
    assert r[7].tolist() == sorted(flood.tolist())[:k]
    want_r, want_s = cref.cosine_topk(_bits(q[[7, 9, 100]]), _bits(m), k, dtype="f16")
    assert np.array_equal(r[[7, 9, 100]].cpu().numpy(), want_r) and np.array_equal(s[[7, 9, 100]].cpu().numpy(), want_s)


def test_gemm_class_scan_full_size_properties_7040_queries_1m_rows():
    """BASELINE configs[3] per-rank size (7,040 queries x 1,048,576 x 768 fp16 rows, top-10): size-independent
    properties instead of an oracle pass - planted self-matches first with score 1, ordered distinct rows, returned
    scores are the exact reference cosines of the returned rows, sharding invariance (top-k(whole) ==
    merge(top-k(two row-interleaved halves)), global row ids), agreement with the exhaustive fp64 kernel on queries of
    the first, a middle and the ragged last query tile."""
    D, M, Q, k = 768, 1_048_576, 7040, 10
    g = torch.Generator(device="cuda").manual_seed(2026)
    from vidmem.memory import EmbeddingMemory, topk_merge
    mem = EmbeddingMemory(M, D, "f16")
    halves = [EmbeddingMemory(M // 2, D, "f16") for _ in range(2)]
    for lo in range(0, M, 262_144):
        x = torch.randn((262_144, D), generator=g, device="cuda", dtype=torch.float32)
        x = (x / x.norm(dim=1, keepdim=True)).to(torch.float16)
        mem.append(x)
        for part in range(2):
            halves[part].append(x[part::2])
    q = torch.randn((Q, D), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
    planted = {0: 3, 255: 4_095, 256: 4_096, 1000: 32_767, 3333: 262_144, 6911: 1_048_575, 7039: 777_777}
    rows_t = mem.rows_tensor()
    for qi, row in planted.items():
        q[qi] = rows_t[row]
    mem.reset_uncertified()
    s, r = mem.topk(q, k)
    assert mem.uncertified_count == 0
    s_np, r_np = s.cpu().numpy(), r.cpu().numpy()
    for qi, row in planted.items():
        assert r_np[qi, 0] == row and abs(s_np[qi, 0] - 1.0) < 1e-12, (qi, row, r_np[qi, 0])
    assert (np.diff(s_np, axis=1) <= 0).all() and (r_np >= 0).all() and (r_np < M).all()
    assert (np.sort(r_np, axis=1)[:, 1:] != np.sort(r_np, axis=1)[:, :-1]).all()
    for qi in (5, 4000, 7038):
        ex = mem.cosine_exact(q[qi:qi + 1], rows_t[r[qi]]).cpu().numpy()
        assert np.array_equal(ex[0], s_np[qi])
    parts = [h.topk(q, k, row_stride=2, row_offset=p) for p, h in enumerate(halves)]
    ms, mr = topk_merge(mem.ctx, torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert np.array_equal(mr.cpu().numpy(), r_np) and np.array_equal(ms.cpu().numpy(), s_np)
    pick = torch.tensor([1, 3500, 6912, 7039], device="cuda")
    s2, r2 = mem.topk(q[pick], k, exact=True)
    assert np.array_equal(r2.cpu().numpy(), r_np[pick.cpu().numpy()])
    assert np.array_equal(s2.cpu().numpy(), s_np[pick.cpu().numpy()])
    for h in halves:
        h.close()
    mem.close()


@pytest.mark.parametrize("k", [32, 58])
def test_large_k_on_a_large_memory_stays_on_the_fast_path(k):
    """k >= 27 keeps 64 candidates per query (KL = 64).  The previous sample-based cut expected ~4 k candidates per
    query in its last pass on memories of a million rows - the capacity of a candidate buffer - so a third of the
    queries overflowed into the exhaustive redo.  The incremental cascade emits ~7 KL per pass whatever the size:
    nothing may be redone here, and the answers equal the exhaustive kernel's."""
    D, M, Q = 256, 1_000_000, 64
    g = torch.Generator(device="cuda").manual_seed(58)
    from vidmem.memory import EmbeddingMemory
    mem = EmbeddingMemory(M, D, "f16")
    for lo in range(0, M, 250_000):
        x = torch.randn((250_000, D), generator=g, device="cuda", dtype=torch.float32)
        mem.append((x / x.norm(dim=1, keepdim=True)).to(torch.float16))
    q = torch.randn((Q, D), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
    mem.reset_uncertified()
    s, r = mem.topk(q, k)
    assert mem.uncertified_count == 0
    s2, r2 = mem.topk(q[:6], k, exact=True)
    assert np.array_equal(r2.cpu().numpy(), r[:6].cpu().numpy()) and np.array_equal(s2.cpu().numpy(), s[:6].cpu().numpy())
    mem.close()


@pytest.mark.parametrize("M,Q", [(150_000, 520), (300_000, 160)], ids=["520q", "160q-one-ragged-tile"])
def test_gemm_class_scan_two_k_tiles(M, Q):
    """D = 128: a score tile of the GEMM-class scan is two K-tiles deep, the shortest stream its tile-boundary logic
    (K-tile 0 staged during the previous tile's last K-tile, K-tile 1 at the boundary) can run on.  520 queries (three
    query tiles, the last one 8 wide) over 150,000 rows / 160 queries (ONE ragged query tile: the smallest query count
    the scan takes) over 300,000 rows; every query against the exhaustive kernel, two dozen against the C oracle."""
    rng = np.random.default_rng(128)
    D, k = 128, 10
    m = torch.tensor(rng.standard_normal((M, D)), dtype=torch.float32).to(torch.float16)
    q = torch.tensor(rng.standard_normal((Q, D)), dtype=torch.float32).to(torch.float16)
    q[:30] = (0.6 * m[rng.integers(0, M, 30)].float() + 0.4 * q[:30].float()).to(torch.float16)
    m[140_000] = m[77]
    q[Q - 1] = m[77]
    mem = _mem("f16", M, D)
    mem.append(m)
    mem.reset_uncertified()
    s, r = mem.topk(q, k)
    assert r[Q - 1, :2].tolist() == [77, 140_000]
    s2, r2 = mem.topk(q, k, exact=True)
    assert np.array_equal(r2.cpu().numpy(), r.cpu().numpy()) and np.array_equal(s2.cpu().numpy(), s.cpu().numpy())
    picks = np.unique(np.concatenate([np.arange(0, Q, 23), [Q - 9, Q - 8, Q - 1]]))
    want_r, want_s = cref.cosine_topk(_bits(q[picks]), _bits(m), k, dtype="f16")
    assert np.array_equal(r[picks].cpu().numpy(), want_r) and np.array_equal(s[picks].cpu().numpy(), want_s)
    assert mem.uncertified_count <= 2


@pytest.mark.parametrize("Q", [64, 600])
def test_newest_rows_form_the_dense_pass(Q):
    """A growing (non-ring) memory whose NEWEST 6,000 rows all resemble the queries - a long scene at the end of a video -
    behind 150,000 unrelated ones.  The cut cascade keeps every score of the newest ~4 k rows first, so its cut starts at
    the scene's level; with the physically first rows as the dense set the last pass found 6,000 rows above the cut of
    the old ones, every candidate buffer overflowed and the whole batch went to the exhaustive redo.  Answers equal the
    exhaustive kernel's either way; what is pinned here is that (almost) nothing is redone.  Emit scan and GEMM-class
    scan."""
    rng = np.random.default_rng(6000 + Q)
    D, M_old, M_new, k = 256, 150_000, 6_000, 10
    old = torch.tensor(rng.standard_normal((M_old, D)), dtype=torch.float32)
    centre = torch.tensor(rng.standard_normal((1, D)), dtype=torch.float32)
    new = centre + 0.6 * torch.tensor(rng.standard_normal((M_new, D)), dtype=torch.float32)
    q = (centre + 0.6 * torch.tensor(rng.standard_normal((Q, D)), dtype=torch.float32)).to(torch.float16)
    mem = _mem("f16", M_old + M_new + 1000, D)
    mem.append(old.to(torch.float16))
    mem.append(new.to(torch.float16))
    mem.reset_uncertified()
    s, r = mem.topk(q, k)
    redone = mem.uncertified_count
    assert (r.cpu().numpy() >= M_old).all()                       # the neighbours are in the scene
    s2, r2 = mem.topk(q, k, exact=True)
    assert np.array_equal(r2.cpu().numpy(), r.cpu().numpy()) and np.array_equal(s2.cpu().numpy(), s.cpu().numpy())
    assert redone <= Q // 50 + 1, redone
