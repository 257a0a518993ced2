"""Developer probe: does running two encoder passes concurrently on two HIP streams raise throughput?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vidmem
from vidmem import specs, synthetic as syn
from vidmem.encoder import FrameEncoder
spec = specs.VIT_B16_224
w = syn.encoder_weights(spec, seed=42)
encs = [FrameEncoder(spec, w, "f16"), FrameEncoder(spec, w, "f16")]
F = int(sys.argv[1]) if len(sys.argv) > 1 else 440
frames = [torch.randint(0, 256, (F, 224, 224, 3), device="cuda", dtype=torch.uint8) for _ in range(2)]
patches = [e.preprocess(f) for e, f in zip(encs, frames)]
def run_single(n):
    for _ in range(n):
        encs[0].encode_patches(patches[0]); encs[0].encode_patches(patches[1])
def run_dual(n, streams):
    for _ in range(n):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                encs[i].encode_patches(patches[i])
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for name, fn in (("single", lambda n: run_single(n)), ("dual", lambda n: run_dual(n, streams))):
    fn(1); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(4); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {2 * F * 4 / dt:.0f} frames/s")
