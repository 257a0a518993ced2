"""CPU: randomized (hypothesis) equivalence of the host-side mirrors with the oracle restatements of the reference's
integer and dict logic - the parts of the path that run on the host in both implementations."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import frames_ref as F
from oracle import similarity_ref as S
from vidmem.extractor import chunk_plan
from vidmem.similarity import merge_batch_similarities


@settings(max_examples=300, deadline=None)
@given(fps=st.sampled_from([10.0, 23.976, 24.0, 25.0, 29.97, 30.0, 59.94, 60.0]),
       total=st.integers(min_value=1, max_value=20_000),
       secs=st.sampled_from([0.5, 1.0, 2.0, 5.0, 7.5, 10.0]),
       fpc=st.integers(min_value=1, max_value=32))
def test_chunk_plan_equals_oracle(fps, total, secs, fpc):
    """src/pipeline/vlm_extractor.py:38-46,57-59,101-107: chunk integers, linspace picks, MM:SS labels."""
    want_total, want = F.chunk_plan_ref(fps, total, secs, fpc)
    got_total, got = chunk_plan(fps, total, secs, fpc)
    assert got_total == want_total
    assert [(p[0], p[1], p[2], p[3], p[4]) for p in got] == [
        (p["chunk_idx"], p["start"], p["end"], p["indices"], p["time"]) for p in want]
    for _, start, end, idx, _ in got:                      # picks stay inside their chunk and never repeat backwards
        assert all(start <= i < end for i in idx) and idx == sorted(idx) and len(idx) <= fpc


ids = st.sampled_from([f"run_{b}_{i}" for b in range(3) for i in range(5)])
scores = st.floats(min_value=-1.0, max_value=1.0, allow_nan=False, width=64)


@settings(max_examples=300, deadline=None)
@given(lists=st.lists(st.lists(st.tuples(ids, scores), max_size=6), max_size=6), k=st.integers(min_value=0, max_value=8))
def test_max_merge_equals_oracle(lists, k):
    """src/components/pre_llm_injector.py:238-249: max per id in first-seen order, stable descending sort, slice."""
    assert merge_batch_similarities(lists, k) == S.merge_batch_similarities_ref(lists, k)


@settings(max_examples=100, deadline=None)
@given(seed=st.integers(min_value=0, max_value=10_000), q=st.integers(1, 4), m=st.integers(1, 40), k=st.integers(1, 12),
       dup=st.booleans(), zero=st.booleans())
def test_numpy_topk_equals_pure_python_loop(seed, q, m, k, dup, zero):
    """The numpy restatement used at mid sizes against the literal Python loop of :346-372 (ties, zero rows)."""
    rng = np.random.default_rng(seed)
    Q = rng.standard_normal((q, 16)).astype(np.float32).astype(np.float64)
    M = rng.standard_normal((m, 16)).astype(np.float32).astype(np.float64)
    if dup and m > 2:
        M[m - 1] = M[0]
    if zero:
        M[m // 2] = 0.0
    rows, sc = S.batch_topk_np(Q, M, k)
    mem = {f"r{i}": M[i].tolist() for i in range(m)}
    want = S.calculate_batch_similarities_ref([r.tolist() for r in Q], mem, k)
    for qi, lst in enumerate(want):
        n = len(lst)
        assert [int(c[1:]) for c, _ in lst] == rows[qi, :n].tolist()
        assert [s for _, s in lst] == sc[qi, :n].tolist()
        assert (rows[qi, n:] == -1).all()


def test_bench_launches_itself_for_n_gt_1():
    """`python bench.py --gpus 2` with no launcher must start two ranks as child processes, rendezvous on 127.0.0.1
    with a free port, relay rank 0's single JSON line and exit 0 (it used to raise SystemExit).  --dry-run
    keeps the ranks on the CPU (gloo) so this runs without a GPU; the measured path differs only after the launch."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3
    # and a world-size mismatch under an external launcher is refused, not silently accepted
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], cwd=root,
                         env=env2, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0
