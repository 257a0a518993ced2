"""Configuration for the drop-in classes: the reference's YAML (config/base_config.yaml, models in
src/core/config.py:5-114) plus two new sections the HIP path needs.

The reference's ``PipelineConfig`` is a pydantic model that silently DROPS unknown keys (``parse_obj`` with the default
``Extra.ignore``; src/core/config.py:97-114), so an ``encoder:`` / ``memory:`` section in the YAML would never reach a
class constructed from it.  ``load_config`` therefore keeps every key of the file (attribute access + ``dict()``, the
two things the reference's classes use) and fills in the defaults below; a plain reference ``PipelineConfig`` object
works too - the two sections then take their defaults (``section()``).

Keys the hot path honours (same names and defaults as the reference where it has them):
  video.chunk_size_seconds = 5.0, video.frames_per_chunk = 5                       src/core/config.py:5-8
  embedder.top_k_chunk_with_batch_similarity = 3, embedder.top_k_similar_batch = 2 src/core/config.py:57-62
  retrieval.top_k = 5, top_k_chunks = 8, compression_threshold = 0.7               src/core/config.py:70-86
New:
  encoder: {arch, dtype, weights, seed, device, top_k, look_ahead_chunks}   which vision encoder stands where the remote VLM was
  memory:  {capacity, ring, dtype, snapshot}               the HBM-resident store that stands where Chunk.embedding was
"""
from __future__ import annotations

import copy
from typing import Any, Dict, Optional

ENCODER_DEFAULTS: Dict[str, Any] = {
    "arch": "vit_b16_224",   # specs.SPECS key (BASELINE.json configs: vit_b16_224, clip_l14_336)
    "dtype": "f16",          # f16 | bf16: GEMM operand and embedding type
    "weights": None,         # .npz of named fp32 arrays (synthetic.encoder_weight_shapes); None = seeded synthetic
    "seed": 42,              # seed of the synthetic weights (no checkpoint can be fetched offline)
    "device": 0,
    "top_k": 5,              # neighbours reported per frame in the output JSON (`similar`)
    "look_ahead_chunks": 0,  # chunks encoded per encoder call (extractor.py: same neighbours and rows as 1, bit for bit);
                             # 0 (default) = as many as fill the encoder's passes within a staging budget; 1 = the
                             # reference's one chunk in flight (src/pipeline/vlm_extractor.py:44-74: a live feed)
}
MEMORY_DEFAULTS: Dict[str, Any] = {
    "capacity": 100_000,     # rows resident in HBM (the reference caps its read-back at 5000: pre_llm_injector.py:398)
    "ring": False,           # True: rolling window, oldest rows overwritten
    "dtype": None,           # None = the encoder's dtype
    "snapshot": None,        # path of an EmbeddingMemory.snapshot to restore from / save to
}
VIDEO_DEFAULTS: Dict[str, Any] = {"chunk_size_seconds": 5.0, "frames_per_chunk": 5, "fps_target": None}
EMBEDDER_DEFAULTS: Dict[str, Any] = {"top_k_chunk_with_batch_similarity": 3, "top_k_similar_batch": 2}
RETRIEVAL_DEFAULTS: Dict[str, Any] = {"top_k": 5, "top_k_chunks": 8, "compression_threshold": 0.7}


class Section:
    """Attribute view of a dict (nested), with the ``dict()`` the reference calls on its config
    (src/pipeline/vlm_extractor.py:82)."""

    def __init__(self, data: Optional[Dict[str, Any]] = None):
        for k, v in (data or {}).items():
            setattr(self, k, Section(v) if isinstance(v, dict) else v)

    def dict(self) -> Dict[str, Any]:
        return {k: (v.dict() if isinstance(v, Section) else copy.deepcopy(v)) for k, v in vars(self).items()}

    def get(self, key: str, default=None):
        return getattr(self, key, default)

    def __repr__(self):
        return f"Section({self.dict()!r})"


def _with_defaults(data: Optional[Dict[str, Any]], defaults: Dict[str, Any]) -> Dict[str, Any]:
    out = dict(defaults)
    out.update(data or {})
    return out


def from_dict(data: Dict[str, Any]) -> Section:
    data = dict(data or {})
    data["video"] = _with_defaults(data.get("video"), VIDEO_DEFAULTS)
    data["encoder"] = _with_defaults(data.get("encoder"), ENCODER_DEFAULTS)
    data["memory"] = _with_defaults(data.get("memory"), MEMORY_DEFAULTS)
    if "embedder" in data:
        data["embedder"] = _with_defaults(data["embedder"], EMBEDDER_DEFAULTS)
    if "retrieval" in data:
        data["retrieval"] = _with_defaults(data["retrieval"], RETRIEVAL_DEFAULTS)
    return Section(data)


def load_config(path: str) -> Section:
    """The reference's ``PipelineConfig.from_yaml`` (src/core/config.py:110-114) without the key filter."""
    import yaml
    with open(path) as f:
        return from_dict(yaml.safe_load(f) or {})


def section(config: Any, name: str, defaults: Dict[str, Any]) -> Section:
    """``config.<name>`` as a Section with defaults filled in, whatever ``config`` is: our Section, the reference's
    pydantic model (no such attribute -> all defaults), a SimpleNamespace or a dict."""
    raw = config.get(name) if isinstance(config, dict) else getattr(config, name, None)
    if raw is None:
        return Section(defaults)
    if isinstance(raw, dict):
        return Section(_with_defaults(raw, defaults))
    if hasattr(raw, "dict") and callable(raw.dict):
        return Section(_with_defaults(raw.dict(), defaults))
    return Section(_with_defaults({k: v for k, v in vars(raw).items()}, defaults))


def config_dict(config: Any) -> Optional[Dict[str, Any]]:
    """What goes under metadata.config in the output JSON (src/pipeline/vlm_extractor.py:82: ``self.config.dict()``)."""
    if hasattr(config, "dict") and callable(config.dict):
        return config.dict()
    if isinstance(config, dict):
        return copy.deepcopy(config)
    try:
        def walk(o):
            if hasattr(o, "__dict__"):
                return {k: walk(v) for k, v in vars(o).items()}
            return o
        return walk(config)
    except Exception:
        return None
