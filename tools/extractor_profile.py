"""Developer probe (GPU box): where the host time of FrameEmbeddingExtractor.process_video goes.
usage: extractor_profile.py [frames=4096] [look_ahead=55] [profile=1]
Same setup as bench.py's `extractor` leg (ViT-B/16 fp16, 16-frame chunks, top-10 + append against 100,000 rows that grow
with the clip); prints frames/s of an unprofiled run, then the cProfile top of a profiled one (main thread only)."""
import asyncio, cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VIDGRAPH_LOG_LEVEL", "WARNING")
import numpy as np, torch
import vidmem  # noqa: F401
from vidmem import specs, synthetic as syn, config as vcfg
from vidmem.encoder import FrameEncoder
from vidmem.memory import EmbeddingMemory
from vidmem.extractor import FrameEmbeddingExtractor

nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
la = int(sys.argv[2]) if len(sys.argv) > 2 else 55
prof = int(sys.argv[3]) if len(sys.argv) > 3 else 1
spec = specs.VIT_B16_224
D, k, R = spec["hidden"], 10, 100_000
enc = FrameEncoder(spec, syn.encoder_weights(spec, seed=42), dtype="f16", device=0)
g = torch.Generator(device="cuda").manual_seed(7)
rows = torch.randn((R, D), generator=g, device="cuda")
rows = (rows / rows.norm(dim=1, keepdim=True)).to(torch.float16)
with tempfile.TemporaryDirectory() as td:
    clip = os.path.join(td, "clip.npy")
    np.save(clip, np.random.default_rng(11).integers(0, 256, size=(nfr, 224, 224, 3), dtype=np.uint8))
    home = os.getcwd()
    os.chdir(td)
    cfg = vcfg.from_dict({"video": {"chunk_size_seconds": 16.0 / 30.0 + 1e-9, "frames_per_chunk": 16},
                          "encoder": {"arch": "vit_b16_224", "dtype": "f16", "top_k": k, "look_ahead_chunks": la},
                          "memory": {"capacity": R + 4 * nfr, "ring": False}})
    mem = EmbeddingMemory(R + 4 * nfr, D, "f16", ring=False, device=0)
    mem.append(rows)
    ex = FrameEmbeddingExtractor(cfg, encoder=enc, memory=mem)
    asyncio.run(ex.process_video(clip, os.path.join(td, "warm.json")))
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        asyncio.run(ex.process_video(clip, os.path.join(td, "out.json")))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{nfr} frames, look-ahead {la}: {nfr / dt:.0f} frames/s ({dt * 1e3:.1f} ms)", flush=True)
    if prof:
        pr = cProfile.Profile()
        pr.enable()
        asyncio.run(ex.process_video(clip, os.path.join(td, "out.json")))
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    os.chdir(home)      # the temporary directory goes away; a profiler's exit hooks want a cwd that exists
