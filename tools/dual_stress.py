"""Developer probe: two-stream encoder mode against the single-stream one, many times (races show up as rare mismatches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vidmem
from vidmem import specs, synthetic as syn
from vidmem.encoder import FrameEncoder
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
spec = dict(specs.VIT_B16_224, layers=3)
w = syn.encoder_weights(spec, seed=17)
plain = FrameEncoder(spec, w, "f16", schedule="one_stream", micro_batch=64)
dual = FrameEncoder(spec, w, "f16", schedule="two_streams", micro_batch=64)
bad = 0
g = torch.Generator(device="cuda").manual_seed(3)
for it in range(iters):
    n = [200, 128, 187, 256][it % 4]
    fr = torch.randint(0, 256, (n, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)
    a = plain.embed_frames(fr); b = dual.embed_frames(fr)
    if not torch.equal(a, b):
        d = (a.float() - b.float()).abs().amax(dim=1)
        rows = (d > 0).nonzero().flatten().tolist()
        print(f"iter {it} n={n}: MISMATCH in {len(rows)} rows, first {rows[:8]}, max diff {float(d.max()):.3e}", flush=True)
        bad += 1
print("mismatching iterations:", bad, "of", iters)
