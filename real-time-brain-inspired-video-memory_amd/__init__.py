"""vidmem: MI355X-native frame-embedding + cosine top-k hot path behind the VidGraph extractor / embedder /
retriever call surfaces (reference: src/pipeline/vlm_extractor.py, src/components/pre_llm_injector.py,
src/pipeline/retriever_hybrid.py).  All arithmetic lives in ``libvidmem.so`` (hand-written HIP for gfx950,
C ABI in include/vidmem.h); this package is the host-side mirror of the reference interfaces."""

__all__ = ["synthetic", "specs"]
__version__ = "0.1.0"
