"""GPU: the drop-in surfaces constructed the way the reference's callers construct them.

  * ``FrameEmbeddingExtractor(config)`` from the pipeline config ALONE (``VLMExtractor(config)``,
    src/pipeline/vlm_extractor.py:19; caller src/cli/main.py:28), its output JSON (:77-88), timing file (:73,:91) and
    the neighbours it reports, checked against the ORACLE (oracle.cref over the embeddings the run stored);
  * ``HipEmbedder.aembed_query / aembed_frames`` - the embedder duck type ``async aembed_query(x) -> List[float]``
    (src/components/neo4j_handler.py:27-31,333; src/components/pre_llm_injector.py:216) - against oracle.vit_ref.
"""
import asyncio
import json
import os

import numpy as np
import pytest
import torch

from oracle import cref
from oracle import frames_ref as F
from oracle import vit_ref as V

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_extractor_from_config_alone_and_neighbours_against_the_oracle(tmp_path, monkeypatch):
    from vidmem import config as C, specs, synthetic as syn
    from vidmem.extractor import FrameEmbeddingExtractor, chunk_plan
    monkeypatch.setitem(specs.SPECS, "vit_b16_2l", dict(specs.VIT_B16_224, layers=2))
    monkeypatch.chdir(tmp_path)                                   # metrics/ and logs/ land where the caller runs
    snap = str(tmp_path / "memory.npz")
    cfg = C.from_dict({
        "video": {"chunk_size_seconds": 1.0, "frames_per_chunk": 6},
        "encoder": {"arch": "vit_b16_2l", "dtype": "f16", "seed": 8, "top_k": 3},
        "memory": {"capacity": 256, "snapshot": snap},
    })
    frames = syn.frames_u8(77, 50, 120, 200)
    clip = tmp_path / "clip.npz"
    np.savez(clip, frames=frames, fps=np.float64(10.0))
    ex = FrameEmbeddingExtractor(cfg)                             # nothing but the config
    out_path = asyncio.run(ex.process_video(str(clip), str(tmp_path / "out.json")))
    out = json.load(open(out_path))
    assert set(out) == {"metadata", "results"}                                            # vlm_extractor.py:77-85
    assert set(out["metadata"]) == {"run_id", "video_path", "total_chunks", "config"}
    assert out["metadata"]["config"]["encoder"]["arch"] == "vit_b16_2l"
    total, plan = chunk_plan(10.0, 50, 1.0, 6)
    assert out["metadata"]["total_chunks"] == total == 5
    stored = ex.memory.rows_tensor().clone()                      # [30, 768] in append order
    assert stored.shape[0] == 30
    run_id = out["metadata"]["run_id"]
    for res, (chunk_idx, start, end, idx, label) in zip(out["results"], plan):
        assert {"time", "content", "chunk_idx", "processing_time"} <= set(res)            # :66-71
        assert res["chunk_idx"] == chunk_idx and res["time"] == label
        first = chunk_idx * 6
        assert res["embedding_rows"] == list(range(first, first + 6))
        if first == 0:
            assert res["similar"] == []
            continue
        want_r, want_s = cref.cosine_topk(_bits(stored[first:first + 6]), _bits(stored[:first]), 3, dtype="f16")
        want = [[[f"{run_id}_{r // 6}_{r % 6}", float(s)] for r, s in zip(rq, sq) if r >= 0]
                for rq, sq in zip(want_r, want_s)]
        assert res["similar"] == want                             # ids by the reference's scheme, fp64 scores bit for bit
    # the stored embeddings themselves against the encoder oracle (first chunk)
    spec = specs.SPECS["vit_b16_2l"]
    px = F.preprocess_ref(frames[plan[0][3]], 224, spec["mean"], spec["std"], layout="chw")
    want_emb = V.vit_forward_ref(spec, syn.encoder_weights(spec, seed=8), px, quant="f16")
    assert rel(stored[:6].float().cpu().numpy(), want_emb) < 1e-3
    # timing file: metrics/vlm_<run_id>.json with one chunk_<i>.vlm_inference entry per chunk (:73,:91)
    m = json.load(open(os.path.join("metrics", f"vlm_{run_id}.json")))
    assert sorted(m["timings"]) == sorted(f"chunk_{i}.vlm_inference" for i in range(5))
    assert set(m["summary"]) == {"total_runtime", "timing_averages", "counts", "batch_count"}
    # memory.snapshot: the run persisted its memory; a second extractor from the same config resumes from it
    ex2 = FrameEmbeddingExtractor(cfg, encoder=ex.encoder)
    assert len(ex2.memory) == 30 and torch.equal(ex2.memory.rows_tensor(), stored)
    assert ex2.memory.id_of(7) == f"{run_id}_1_1"


@pytest.mark.parametrize("memory_cfg", [{"capacity": 48, "ring": True}, {"capacity": 4096, "ring": False},
                                        {"capacity": 80, "ring": True}],
                         ids=["ring-wraps", "grows", "ring-wraps-late"])
def test_look_ahead_groups_change_nothing_but_the_time(tmp_path, monkeypatch, memory_cfg):
    """``encoder.look_ahead_chunks: N`` encodes N chunks per encoder call; every chunk still sees the memory as it
    stood after the chunk before it (src/pipeline/vlm_extractor.py:44-74 order) - by its own top-k and append when a
    ring wraps inside the group, by ONE search per group (memory before the group + the group's earlier chunks through
    the exact all-pairs kernel, merged) when nothing can be overwritten meanwhile.  Stored rows, row numbers and
    reported neighbours (ids AND fp64 scores) must equal the N = 1 run exactly, with a ragged last group, a ragged last
    chunk, an empty memory at the start, and a ring that wraps from the first / from a later group on."""
    from vidmem import config as C, specs, synthetic as syn
    from vidmem.extractor import FrameEmbeddingExtractor
    monkeypatch.setitem(specs.SPECS, "vit_b16_2l", dict(specs.VIT_B16_224, layers=2))
    monkeypatch.chdir(tmp_path)
    frames = syn.frames_u8(5, 95, 96, 128)
    clip = tmp_path / "clip.npz"
    np.savez(clip, frames=frames, fps=np.float64(10.0))           # 9 chunks of 10 frames, 7 picked per chunk; 5 left over
    runs = {}
    enc = None
    for n in (1, 4, 0):      # 0 = sized from the encoder's pass (here: more chunks than the clip has, one ramp of groups)
        cfg = C.from_dict({
            "video": {"chunk_size_seconds": 1.0, "frames_per_chunk": 7},
            "encoder": {"arch": "vit_b16_2l", "dtype": "f16", "seed": 3, "top_k": 4, "look_ahead_chunks": n},
            "memory": memory_cfg,
        })
        ex = FrameEmbeddingExtractor(cfg, encoder=enc)
        enc = ex.encoder
        out = json.load(open(asyncio.run(ex.process_video(str(clip), str(tmp_path / f"out{n}.json")))))
        rid = out["metadata"]["run_id"]
        res = [{**r, "processing_time": None, "group_time": None, "group_chunks": None,   # the time is what may change
                "similar": [[(i.replace(rid, "RUN") if i else i, s) for i, s in fr] for fr in r["similar"]]}
               for r in out["results"]]
        runs[n] = (res, ex.memory.rows_tensor().clone(), len(ex.memory))
    assert len(runs[1][0]) == 9 and runs[1][2] == 63
    assert runs[1][0] == runs[4][0]
    assert torch.equal(runs[1][1], runs[4][1]) and runs[1][2] == runs[4][2]
    assert runs[1][0] == runs[0][0] and torch.equal(runs[1][1], runs[0][1]) and runs[1][2] == runs[0][2]
    assert any(fr for r in runs[4][0] for fr in r["similar"])


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_hip_embedder_surface_against_the_encoder_oracle(dtype):
    from vidmem import specs, synthetic as syn
    from vidmem.encoder import FrameEncoder, HipEmbedder
    spec = dict(specs.VIT_B16_224, layers=2)
    w = syn.encoder_weights(spec, seed=13)
    emb = HipEmbedder(FrameEncoder(spec, w, dtype))
    frames = syn.frames_u8(31, 5, 224, 224)
    px = F.preprocess_ref(frames, 224, spec["mean"], spec["std"], layout="chw")
    want = V.vit_forward_ref(spec, w, px, quant=dtype)
    many = asyncio.run(emb.aembed_frames(frames))                 # List[List[float]], one launch
    assert isinstance(many, list) and len(many) == 5 and all(isinstance(v, list) and len(v) == 768 for v in many)
    assert all(isinstance(x, float) for x in many[0])
    # bf16: 1.5 x the operand-only floor of this 2-layer stack (tests/golden/bf16_floor.py; test_encoder_gpu.py bars)
    bar = 1e-3 if dtype == "f16" else max(1e-3, 1.5 * rel(
        V.vit_forward_ref(spec, w, px, quant={p: "bf16" for p in V.OPERAND_POINTS}),
        V.vit_forward_ref(spec, w, px, quant=None)))
    assert rel(np.asarray(many), want) < bar
    one = asyncio.run(emb.aembed_query(frames[2]))                # one frame, uint8 [H,W,3] BGR -> List[float]
    assert isinstance(one, list) and len(one) == 768
    assert one == many[2]                                         # batching is invisible, value for value
    t = emb.embed_frames_tensor(torch.from_numpy(frames).cuda())
    assert t.shape == (5, 768) and np.array_equal(t.float().cpu().numpy(), np.asarray(many, np.float32))
