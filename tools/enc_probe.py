"""Developer probe: N encoder passes over one micro-batch of synthetic frames (for rocprofv3 --pmc runs)."""
import sys, torch
sys.path.insert(0, ".")
import vidmem  # noqa
from vidmem import specs, synthetic as syn
from vidmem.encoder import FrameEncoder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 441
spec = specs.VIT_B16_224
enc = FrameEncoder(spec, syn.encoder_weights(spec, seed=42), dtype="f16", device=0)
g = torch.Generator(device="cuda").manual_seed(1)
fr = torch.randint(0, 256, (B, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)
for _ in range(n):
    e = enc.embed_frames(fr)
torch.cuda.synchronize()
print("ok", float(e.float().abs().sum()))
