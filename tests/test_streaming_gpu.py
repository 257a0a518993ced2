"""GPU: persistence round trip and the hipGraph-captured streaming loop (BASELINE config C5 shape, small)."""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


def test_snapshot_restore_roundtrip(tmp_path):
    from vidmem.memory import EmbeddingMemory
    rng = np.random.default_rng(2)
    rows = torch.tensor(rng.standard_normal((150, 256)), dtype=torch.float32).to(torch.float16)
    mem = EmbeddingMemory(100, 256, "f16", ring=True, graph_uuid="g-1")
    mem.append(rows[:90], ids=[f"u_{i}" for i in range(90)])
    mem.append(rows[90:], ids=[f"u_{i}" for i in range(90, 150)])      # wraps: rows 50..149 survive
    p = str(tmp_path / "mem.npz")
    mem.snapshot(p)
    back = EmbeddingMemory.restore(p)
    assert len(back) == 100 and back.graph_uuid == "g-1" and back.ids[0] == "u_50" and back.ids[-1] == "u_149"
    assert torch.equal(back.rows_tensor().cpu(), rows[50:])
    q = rows[[60, 149]]
    s1, r1 = mem.topk(q, 4)
    s2, r2 = back.topk(q, 4)
    assert torch.equal(s1, s2) and torch.equal(r1 - 50, r2)   # same answers, row ids rebased to 0


def test_streaming_graph_matches_eager_and_oracle():
    from vidmem import specs, synthetic as syn
    from vidmem.encoder import FrameEncoder
    from vidmem.memory import EmbeddingMemory
    from vidmem.streaming import StreamingSession
    spec = dict(specs.VIT_B16_224, layers=1)
    w = syn.encoder_weights(spec, seed=5)
    enc = FrameEncoder(spec, w, "f16")
    B, cap, k = 16, 64, 5
    mem = EmbeddingMemory(cap, 768, "f16", ring=True)
    seed_rows = torch.from_numpy(syn.unit_rows(3, "seed", 40, 768)).to(torch.float16)
    mem.append(seed_rows)
    sess = StreamingSession(enc, mem, B, 360, 640, top_k=k, warmup=1)
    assert len(mem) == 40 and sess.rows_appended == 40      # warm-up and capture leave the user's memory untouched
    hist = [seed_rows.cuda()]
    for step in range(4):                  # crosses the ring wrap
        frames = torch.from_numpy(syn.frames_u8(100 + step, B, 360, 640)).cuda()
        emb, scores, rows = sess.push(frames)
        torch.cuda.synchronize()
        want_emb = enc.embed_frames(frames)
        assert torch.equal(emb, want_emb)
        allrows = torch.cat(hist)
        lo = max(0, allrows.shape[0] - cap)
        want_r, want_s = cref.cosine_topk(_bits(emb), _bits(allrows[lo:]), k, dtype="f16")
        want_r = np.where(want_r >= 0, want_r + lo, -1)
        assert sess.uncertified_last_push == 0
        assert np.array_equal(rows.cpu().numpy(), want_r)
        assert np.array_equal(scores.cpu().numpy(), want_s)
        hist.append(emb.clone())
    assert sess.rows_appended == 40 + 4 * B
    # host mirror lags the replays until sync(); afterwards eager calls (exhaustive search, append) line up again
    assert len(mem) == 40
    assert sess.sync() == 40 + 4 * B and len(mem) == 40 + 4 * B and len(mem.ids) == len(mem)
    allrows = torch.cat(hist)
    q = allrows[[45, 103]]
    s_e, r_e = mem.topk(q, 3, exact=True)
    want_r, want_s = cref.cosine_topk(_bits(q), _bits(allrows[-cap:]), 3, dtype="f16")
    assert np.array_equal(r_e.cpu().numpy(), want_r + (allrows.shape[0] - cap))
    assert np.array_equal(s_e.cpu().numpy(), want_s)
    first = mem.append(seed_rows[:2], ids=["a", "b"])
    assert first == 40 + 4 * B and mem.id_of(first + 1) == "b"


def test_streaming_redoes_uncertified_queries_inside_the_replay():
    """A chunk of IDENTICAL frames appends 16 exact duplicates per push: after three pushes a query has more exact
    ties than candidate slots (KL = 16 at k = 10), the scan cannot certify it, and the graph's own redo kernel must
    return the reference answer (lowest row ids first) - per-push counter > 0, results equal to the C oracle."""
    from vidmem import specs, synthetic as syn
    from vidmem.encoder import FrameEncoder
    from vidmem.memory import EmbeddingMemory
    from vidmem.streaming import StreamingSession
    spec = dict(specs.VIT_B16_224, layers=1)
    enc = FrameEncoder(spec, syn.encoder_weights(spec, seed=5), "f16")
    B, cap, k = 16, 256, 10
    mem = EmbeddingMemory(cap, 768, "f16", ring=True)
    seed_rows = torch.from_numpy(syn.unit_rows(3, "seed", 40, 768)).to(torch.float16)
    mem.append(seed_rows)
    sess = StreamingSession(enc, mem, B, 224, 224, top_k=k, warmup=1)
    one = torch.from_numpy(syn.frames_u8(7, 1, 224, 224)).cuda()
    frames = one.expand(B, -1, -1, -1).contiguous()
    hist = [seed_rows.cuda()]
    counts = []
    for step in range(4):
        emb, scores, rows = sess.push(frames)
        counts.append(sess.uncertified_last_push)
        allrows = torch.cat(hist)
        want_r, want_s = cref.cosine_topk(_bits(emb), _bits(allrows), k, dtype="f16")
        assert np.array_equal(rows.cpu().numpy(), want_r), step
        assert np.array_equal(scores.cpu().numpy(), want_s), step
        hist.append(emb.clone())
    assert counts[0] == 0 and counts[-1] == B      # 48 stored duplicates > 16 candidate slots: every query redone


def test_session_survives_larger_eager_calls_on_the_same_encoder_and_memory():
    """The captured graph writes through the SESSION's workspaces: a larger eager batch that re-allocates the
    encoder's / memory's shared scratch must not disturb later replays (stale-pointer use-after-free)."""
    from vidmem import specs, synthetic as syn
    from vidmem.encoder import FrameEncoder
    from vidmem.memory import EmbeddingMemory
    from vidmem.streaming import StreamingSession
    spec = dict(specs.VIT_B16_224, layers=1)
    enc = FrameEncoder(spec, syn.encoder_weights(spec, seed=5), "f16")
    B, cap, k = 16, 512, 5
    mem = EmbeddingMemory(cap, 768, "f16", ring=True)
    seed_rows = torch.from_numpy(syn.unit_rows(3, "seed", 100, 768)).to(torch.float16)
    mem.append(seed_rows)
    sess = StreamingSession(enc, mem, B, 224, 224, top_k=k, warmup=1)
    f0 = torch.from_numpy(syn.frames_u8(200, B, 224, 224)).cuda()
    res0 = sess.push(f0)
    torch.cuda.synchronize()                          # the replay runs on the session's stream
    emb0, s0, r0 = [t.clone() for t in res0]
    big = torch.from_numpy(syn.frames_u8(201, 96, 224, 224)).cuda()
    e_big = enc.embed_frames(big)                     # grows the encoder's shared workspace
    mem.topk(e_big, 40)                               # grows the memory's shared top-k scratch
    junk = [torch.full((1 << 22,), 0x7F, dtype=torch.uint8, device="cuda") for _ in range(8)]   # reuse freed blocks
    f1 = torch.from_numpy(syn.frames_u8(202, B, 224, 224)).cuda()
    emb1, s1, r1 = sess.push(f1)
    torch.cuda.synchronize()
    assert torch.equal(emb1, enc.embed_frames(f1))
    allrows = torch.cat([seed_rows.cuda(), emb0])
    want_r, want_s = cref.cosine_topk(_bits(emb1), _bits(allrows), k, dtype="f16")
    assert np.array_equal(r1.cpu().numpy(), want_r) and np.array_equal(s1.cpu().numpy(), want_s)
    del junk
