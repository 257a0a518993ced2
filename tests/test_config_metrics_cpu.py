"""CPU: the configuration and timing-file shapes of the drop-in classes against the reference's own files
(config/base_config.yaml is read where it lies under /root/reference when present; the metrics JSON shape restates
src/core/metrics.py:9-66)."""
import json
import os
import textwrap
from types import SimpleNamespace

import numpy as np
import pytest

REF_YAML = "/root/reference/config/base_config.yaml"


def test_load_config_keeps_reference_keys_and_adds_sections(tmp_path):
    from vidmem import config as C
    y = tmp_path / "cfg.yaml"
    y.write_text(textwrap.dedent("""
        video: {chunk_size_seconds: 2.5, frames_per_chunk: 16}
        embedder: {endpoint: "http://x", api_key: "", model: "m", top_k_similar_batch: 4}
        retrieval: {top_k_chunks: 10}
        encoder: {arch: clip_l14_336, dtype: bf16}
        memory: {capacity: 4096, ring: true}
        something_else: {kept: 1}
    """))
    cfg = C.load_config(str(y))
    assert cfg.video.chunk_size_seconds == 2.5 and cfg.video.frames_per_chunk == 16 and cfg.video.fps_target is None
    assert cfg.embedder.top_k_similar_batch == 4 and cfg.embedder.top_k_chunk_with_batch_similarity == 3   # default :61
    assert cfg.retrieval.top_k_chunks == 10 and cfg.retrieval.compression_threshold == 0.7                  # default :81
    assert cfg.encoder.arch == "clip_l14_336" and cfg.encoder.dtype == "bf16" and cfg.encoder.seed == 42
    assert cfg.memory.capacity == 4096 and cfg.memory.ring is True and cfg.memory.dtype is None
    d = cfg.dict()
    assert d["something_else"] == {"kept": 1} and d["encoder"]["top_k"] == 5
    json.dumps(d)  # what process_video writes under metadata.config


@pytest.mark.skipif(not os.path.exists(REF_YAML), reason="reference tree not present (GPU box)")
def test_reference_yaml_loads_with_defaults_for_the_new_sections():
    from vidmem import config as C
    cfg = C.load_config(REF_YAML)
    assert cfg.video.chunk_size_seconds == 5.0 and cfg.video.frames_per_chunk == 5          # base_config.yaml:1-4
    assert cfg.embedder.top_k_chunk_with_batch_similarity == 3 and cfg.embedder.top_k_similar_batch == 2
    assert cfg.retrieval.top_k_chunks == 3 and cfg.retrieval.compression_threshold == 0.15
    assert cfg.encoder.dict() == C.ENCODER_DEFAULTS and cfg.memory.dict() == C.MEMORY_DEFAULTS


def test_section_reads_any_config_shape():
    from vidmem import config as C
    ns = SimpleNamespace(video=SimpleNamespace(chunk_size_seconds=1.0, frames_per_chunk=6))
    assert C.section(ns, "video", C.VIDEO_DEFAULTS).frames_per_chunk == 6
    assert C.section(ns, "encoder", C.ENCODER_DEFAULTS).arch == "vit_b16_224"          # absent -> defaults

    class Pyd:                                              # a pydantic-model-shaped object: .dict() on sections
        class video:
            @staticmethod
            def dict():
                return {"chunk_size_seconds": 3.0, "frames_per_chunk": 2, "fps_target": None}
    assert C.section(Pyd, "video", C.VIDEO_DEFAULTS).chunk_size_seconds == 3.0
    assert C.section({"memory": {"capacity": 7}}, "memory", C.MEMORY_DEFAULTS).capacity == 7


def test_metrics_file_has_the_reference_shape(tmp_path):
    from vidmem.metrics import MetricsTracker
    m = MetricsTracker()
    m.record_timing("chunk_0", "vlm_inference", 0.25)      # the key src/pipeline/vlm_extractor.py:73 writes
    m.record_timing("chunk_0", "vlm_inference", 0.75)
    m.record_timing("chunk_1", "vlm_inference", 0.5)
    m.record_count("frames", "embedded", 16)
    m.record_count("frames", "embedded", 16)
    m.add_batch_metrics({"batch": 0})
    p = tmp_path / "metrics" / "vlm_x.json"
    m.save_metrics(str(p))
    d = json.load(open(p))
    assert set(d) == {"start_time", "timings", "counts", "batch_metrics", "end_time", "summary"}   # metrics.py:11-16,60-61
    assert d["timings"] == {"chunk_0.vlm_inference": [0.25, 0.75], "chunk_1.vlm_inference": [0.5]}
    assert d["counts"] == {"frames.embedded": 32}
    assert set(d["summary"]) == {"total_runtime", "timing_averages", "counts", "batch_count"}      # metrics.py:40-54
    assert d["summary"]["timing_averages"]["chunk_0.vlm_inference"] == 0.5 and d["summary"]["batch_count"] == 1


def test_logger_honours_vidgraph_log_level(monkeypatch, tmp_path):
    import logging
    from vidmem.metrics import get_logger
    monkeypatch.setenv("VIDGRAPH_LOG_LEVEL", "debug")
    monkeypatch.setenv("VIDGRAPH_LOG_DIR", str(tmp_path / "logs"))
    lg = get_logger("vidmem.test_logger_cpu")
    assert lg.level == logging.DEBUG and (tmp_path / "logs" / "vidmem.test_logger_cpu.log").exists()


def test_tall_frames_are_not_downscaled_on_the_host():
    """The ONE deliberate deviation from src/pipeline/vlm_extractor.py:114-116 (extractor.py docstring): frames taller
    than 720 rows reach the preprocess kernel at their source size."""
    from vidmem.extractor import FrameEmbeddingExtractor, _ArraySource
    frames = np.zeros((3, 1080, 16, 3), np.uint8)
    ex = FrameEmbeddingExtractor.__new__(FrameEmbeddingExtractor)      # no GPU objects needed for the host pick
    got = ex._read_chunk(_ArraySource(frames, 30.0), [0, 2, 7])
    assert len(got) == 2 and all(f.shape == (1080, 16, 3) for f in got)   # index 7 unreadable -> dropped (:111-112)
