"""CPU: frame-side oracle (chunk plan restating src/pipeline/vlm_extractor.py:38-59,101-107; build-defined preprocess)
and the host-side mirror of the same integers in vidmem.extractor."""
import json

import numpy as np
import pytest
import torch

from oracle import frames_ref as F

# (fps, total_frames, chunk_size_seconds, frames_per_chunk) -> expected, worked by hand from the reference's lines
TABLE = [
    ((30.0, 900, 5.0, 5), dict(total_chunks=6, first=(0, 150, [0, 37, 74, 111, 149], "00:00-00:05"),
                               last=(750, 900, [750, 787, 824, 861, 899], "00:25-00:30"))),
    ((29.97, 1000, 5.0, 5), dict(total_chunks=6, first=(0, 149, [0, 37, 74, 111, 148], "00:00-00:04"),
                                 last=(745, 894, [745, 782, 819, 856, 893], "00:24-00:29"))),
    ((25.0, 60, 5.0, 16), dict(total_chunks=1, first=(0, 60, None, "00:00-00:02"), last=None)),  # max(1, int(60/125))
    ((10.0, 35, 1.0, 16), dict(total_chunks=3, first=(0, 10, list(range(10)), "00:00-00:01"),
                               last=(20, 30, list(range(20, 30)), "00:02-00:03"))),
]


@pytest.mark.parametrize("args,want", TABLE)
def test_chunk_plan_matches_reference_arithmetic(args, want):
    total, plan = F.chunk_plan_ref(*args)
    assert total == want["total_chunks"] and len(plan) == total
    s, e, idx, label = want["first"]
    assert (plan[0]["start"], plan[0]["end"], plan[0]["time"]) == (s, e, label)
    if idx is not None:
        assert plan[0]["indices"] == idx
    else:  # 60 frames < chunk of 125: one chunk of min(16, 60) picks over [0, 59]
        assert plan[0]["indices"] == [int(i) for i in np.linspace(0, 59, 16, dtype=int)]
    if want["last"]:
        s, e, idx, label = want["last"]
        assert (plan[-1]["start"], plan[-1]["end"], plan[-1]["indices"], plan[-1]["time"]) == (s, e, idx, label)
    # host-side mirror used by FrameEmbeddingExtractor
    from vidmem.extractor import chunk_plan
    total2, plan2 = chunk_plan(*args)
    assert total2 == total
    assert [(p[0], p[1], p[2], p[3], p[4]) for p in plan2] == [
        (p["chunk_idx"], p["start"], p["end"], p["indices"], p["time"]) for p in plan]


def test_downscale_size():
    assert F.downscale_720_size_ref(1080, 1920) == (720, 1280)
    assert F.downscale_720_size_ref(720, 1280) == (720, 1280)
    assert F.downscale_720_size_ref(2160, 3840) == (720, 1280)
    assert F.downscale_720_size_ref(1081, 1921) == (720, int(1921 * (720 / 1081)))


@pytest.mark.parametrize("hw", [(224, 224), (360, 640), (97, 131), (1080, 1920)])
def test_preprocess_ref_equals_torch_bilinear(hw):
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, size=(2, hw[0], hw[1], 3), dtype=np.uint8)
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    got = F.preprocess_ref(frames, 224, mean, std, layout="chw")
    rgb = torch.from_numpy(frames[..., ::-1].copy()).permute(0, 3, 1, 2).float()
    t = torch.nn.functional.interpolate(rgb, size=(224, 224), mode="bilinear", align_corners=False)
    want = (t / 255.0 - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    assert np.abs(got - want.numpy()).max() < 5e-4  # same taps; fp32 rounding order differs (0.03 grey levels)


def test_preprocess_ref_patch_layout_is_a_permutation():
    rng = np.random.default_rng(4)
    frames = rng.integers(0, 256, size=(1, 64, 64, 3), dtype=np.uint8)
    chw = F.preprocess_ref(frames, 56, (0.5,) * 3, (0.5,) * 3, layout="chw")
    pat = F.preprocess_ref(frames, 56, (0.5,) * 3, (0.5,) * 3, layout="patches", patch=14, k_pad=640)
    assert pat.shape == (1, 16, 640) and (pat[..., 588:] == 0).all()
    # patch (gy, gx), k = c*196 + py*14 + px
    assert pat[0, 5, 2 * 196 + 3 * 14 + 7] == chw[0, 2, 1 * 14 + 3, 1 * 14 + 7]


def test_extractor_sources_and_json_shape(tmp_path, monkeypatch):
    """Array source + JSON layout of FrameEmbeddingExtractor without touching the GPU (encoder/memory stubbed)."""
    monkeypatch.chdir(tmp_path)  # metrics/vlm_<run>.json and logs/ land where the caller runs, not in the repo
    import asyncio
    from types import SimpleNamespace
    from vidmem import extractor as X

    class FakeEnc:
        device = torch.device("cpu")

        def embed_frames(self, frames):
            return torch.zeros((frames.shape[0], 8))

    class FakeMem:
        searchable = 0
        def __init__(self): self.n = 0
        def append(self, emb, ids=None, meta=None):
            first = self.n; self.n += emb.shape[0]; return first
        def id_of(self, r): return None

    class HostStager:  # ingest.FrameStager's interface without a device
        def __init__(self, n, h, w, device): self.shape, self.staged = (n, h, w, 3), 0
        def stage(self, fr): self.staged += 1; return torch.from_numpy(np.stack(fr))
        def get(self, ticket): return ticket
        def done(self, ticket): pass

    frames = np.zeros((35, 8, 8, 3), np.uint8)
    p = tmp_path / "clip.npz"
    np.savez(p, frames=frames, fps=np.float64(10.0))
    cfg = SimpleNamespace(video=SimpleNamespace(chunk_size_seconds=1.0, frames_per_chunk=16))
    ex = X.FrameEmbeddingExtractor(cfg, FakeEnc(), FakeMem(), top_k=3, stager_factory=HostStager)
    orig_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        out = asyncio.run(ex.process_video(str(p), str(tmp_path / "out.json")))
    finally:
        torch.cuda.synchronize = orig_sync
    d = json.load(open(out))
    assert set(d) == {"metadata", "results"} and d["metadata"]["total_chunks"] == 3
    assert [r["chunk_idx"] for r in d["results"]] == [0, 1, 2]
    assert set(d["results"][0]) >= {"time", "content", "chunk_idx", "processing_time"}  # vlm_extractor.py:66-71
    assert d["results"][2]["embedding_rows"] == list(range(20, 30)) and d["results"][1]["time"] == "00:01-00:02"
    assert ex._stager.staged == 3 and ex._stager.shape == (16, 8, 8, 3)   # one staging per chunk, sized once


def test_a_failing_group_stops_and_joins_the_staging_thread(tmp_path, monkeypatch):
    """process_video reads and stages frames on a thread of its own, ahead of the launches.  When the main loop raises
    (here: the encoder stand-in fails on the second group; in production a full non-ring memory, a top-k error, a failing
    consumer) that thread must be stopped and JOINED before the source is released - a cv2.VideoCapture released during
    a read on another thread can use freed memory - and no thread may be left behind."""
    monkeypatch.chdir(tmp_path)
    import asyncio
    import threading
    import time
    from types import SimpleNamespace
    from vidmem import extractor as X
    events = []

    class SlowSource(X._ArraySource):
        def read_many(self, picks):
            events.append(("read", threading.current_thread().name))
            time.sleep(0.05)                       # the staging thread is inside a read when the consumer fails
            return super().read_many(picks)

        def release(self):
            events.append(("release", threading.current_thread().name,
                           [t.name for t in threading.enumerate() if t.name == "vidmem-stage"]))

    class FailingEnc:
        device = torch.device("cpu")
        calls = 0

        def embed_frames(self, frames):
            FailingEnc.calls += 1
            if FailingEnc.calls == 2:
                raise RuntimeError("encoder failed on group 2")
            return torch.zeros((frames.shape[0], 8))

    class FakeMem:
        searchable, ring, capacity = 0, False, 10 ** 9
        def __init__(self): self.n = 0
        def __len__(self): return self.n
        def append(self, emb, ids=None, meta=None):
            first = self.n; self.n += emb.shape[0]; return first
        def id_of(self, r): return None

    class HostStager:
        def __init__(self, n, h, w, device): self.shape = (n, h, w, 3)
        def stage(self, fr): return torch.from_numpy(np.ascontiguousarray(fr))
        def get(self, ticket): return ticket
        def done(self, ticket): pass

    frames = np.zeros((16 * 12, 4, 4, 3), np.uint8)
    monkeypatch.setattr(X, "open_source", lambda path: SlowSource(frames, 30.0))
    cfg = SimpleNamespace(video=SimpleNamespace(chunk_size_seconds=16 / 30 + 1e-9, frames_per_chunk=16),
                          encoder=SimpleNamespace(look_ahead_chunks=2, top_k=0))
    ex = X.FrameEmbeddingExtractor(cfg, FailingEnc(), FakeMem(), top_k=0, stager_factory=HostStager)
    before = threading.active_count()
    with pytest.raises(RuntimeError, match="group 2"):
        asyncio.run(ex.process_video("clip.fake", str(tmp_path / "out.json")))
    assert threading.active_count() == before
    rel = [e for e in events if e[0] == "release"]
    assert len(rel) == 1 and rel[0][2] == [], "the source was released while the staging thread was alive"
    assert all(e[1] == "vidmem-stage" for e in events if e[0] == "read")     # only that thread touches the source
    # ... and the extractor is usable again afterwards
    FailingEnc.calls = 10
    out = asyncio.run(ex.process_video("clip.fake", str(tmp_path / "out.json")))
    assert len(json.load(open(out))["results"]) == 12


def test_cv2_source_branch_with_a_stand_in_module(tmp_path, monkeypatch):
    """``open_source`` on a real video path goes through ``_Cv2Source`` (cv2.VideoCapture + CAP_PROP_POS_FRAMES seeks,
    src/pipeline/vlm_extractor.py:32-39,108-111).  OpenCV is not installed in the build image, so a stand-in module with
    the four calls the class makes is injected: frame count / fps come from the capture, ``read(idx)`` seeks then reads,
    an unreadable frame is ``None`` (dropped by the chunk reader, :111), ``release`` reaches the capture."""
    import sys
    import types
    from vidmem import extractor as X
    frames = np.arange(7 * 4 * 6 * 3, dtype=np.uint8).reshape(7, 4, 6, 3)
    log = []

    class FakeCapture:
        def __init__(self, path):
            log.append(("open", path))
            self.pos = 0

        def get(self, prop):
            return {5: 25.0, 7: float(len(frames)), 3: 6.0, 4: 4.0}[prop]

        def set(self, prop, value):
            assert prop == 1
            self.pos = int(value)

        def read(self):
            ok = 0 <= self.pos < len(frames) and self.pos != 3        # frame 3 is unreadable
            return (ok, frames[self.pos].copy() if ok else None)

        def release(self):
            log.append(("release",))

    fake = types.SimpleNamespace(VideoCapture=FakeCapture, CAP_PROP_POS_FRAMES=1, CAP_PROP_FPS=5, CAP_PROP_FRAME_COUNT=7,
                                 CAP_PROP_FRAME_WIDTH=3, CAP_PROP_FRAME_HEIGHT=4)
    monkeypatch.setitem(sys.modules, "cv2", fake)
    src = X.open_source(str(tmp_path / "clip.mp4"))
    assert isinstance(src, X._Cv2Source) and src.fps == 25.0 and src.total == 7 and src.frame_hw == (4, 6)
    assert np.array_equal(src.read(5), frames[5]) and src.read(3) is None and src.read(9) is None
    ex = X.FrameEmbeddingExtractor.__new__(X.FrameEmbeddingExtractor)     # only the chunk reader is exercised
    got = ex._read_chunk(src, [0, 3, 6])
    assert len(got) == 2 and np.array_equal(got[1], frames[6])             # the unreadable pick is dropped
    src.release()
    assert log == [("open", str(tmp_path / "clip.mp4")), ("release",)]


def test_array_source_group_reads(tmp_path):
    """``_ArraySource.read_many`` (the slab path of the extractor's staging thread): a view of the clip for consecutive
    picks, a gathered copy for the reference's ``np.linspace`` picks with gaps (src/pipeline/vlm_extractor.py:107);
    either way the frames ``read`` returns one by one."""
    from vidmem import extractor as X
    frames = np.random.default_rng(6).integers(0, 256, size=(40, 6, 5, 3), dtype=np.uint8)
    p = tmp_path / "clip.npy"
    np.save(p, frames)
    src = X.open_source(str(p))
    assert src.total == 40 and src.fps == 30.0
    run = src.read_many(list(range(7, 23)))
    assert run.shape == (16, 6, 5, 3) and np.shares_memory(run, src.frames) and (run == frames[7:23]).all()
    picks = [int(i) for i in np.linspace(3, 38, 8, dtype=int)]
    got = src.read_many(picks)
    assert (got == np.stack([src.read(i) for i in picks])).all() and not np.shares_memory(got, src.frames)
    assert src.read(40) is None and src.read(-1) is None
    # the plan's picks of a whole group, chunk after chunk, are one run when every frame of a chunk is picked
    _n, plan = X.chunk_plan(30.0, 40, 16 / 30 + 1e-9, 16)
    allp = [i for (_c, _s, _e, idx, _t) in plan for i in idx]
    assert allp == list(range(32)) and np.shares_memory(src.read_many(allp), src.frames)
    # look_ahead_chunks: 0 sizes the groups from the encoder's pass and a staging budget
    ex = X.FrameEmbeddingExtractor.__new__(X.FrameEmbeddingExtractor)
    ex.encoder = type("E", (), {"micro_batch": staticmethod(lambda b: min(b, 883))})()
    assert src.frame_hw == (6, 5) and ex._auto_look_ahead(src, 16) == 55
    assert ex._auto_look_ahead(src, 16, slot_bytes=16 * 6 * 5 * 3 * 7 + 1) == 7        # the budget binds
    big = type("S", (), {"frame_hw": (1080, 1920)})()
    assert ex._auto_look_ahead(big, 16) == 3                                           # 16 x 6.2 MB per chunk
    ex.encoder = type("E", (), {"micro_batch": staticmethod(lambda b: min(b, 883)), "schedule": "auto"})()
    assert ex._auto_look_ahead(src, 16) == 110                                         # two passes side by side
    ex.encoder = object()
    assert ex._auto_look_ahead(src, 16) == 1 and ex._auto_look_ahead(type("S", (), {})(), 16) == 1
