"""Developer probe: one 880-frame ViT-B/16 encode, eager launches against a captured hipGraph replay (round 2: 34.1 ms
eager, 34.5 ms replayed - the ~180 kernels of a pass are long enough that launch gaps do not show)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, vidmem
from vidmem import specs, synthetic as syn
from vidmem.encoder import FrameEncoder
spec = specs.VIT_B16_224
enc = FrameEncoder(spec, syn.encoder_weights(spec, seed=42), dtype="f16")
F = 880
frames = torch.randint(0, 256, (F, 224, 224, 3), device="cuda", dtype=torch.uint8)
for _ in range(3): e = enc.embed_frames(frames)
torch.cuda.synchronize()
def timeit(fn, n=8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager ms", timeit(lambda: enc.embed_frames(frames)))
ws = enc.new_workspace(F)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): out = enc.embed_frames(frames, workspace=ws)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = enc.embed_frames(frames, workspace=ws)
torch.cuda.synchronize()
print("graph ms", timeit(lambda: g.replay()))
print("eager ms", timeit(lambda: enc.embed_frames(frames)))
print("graph ms", timeit(lambda: g.replay()))
print("equal", torch.equal(out, enc.embed_frames(frames)))
