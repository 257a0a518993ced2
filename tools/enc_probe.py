"""Developer probe: encoder passes over synthetic frames with the per-kernel HIP-event breakdown.
   python tools/enc_probe.py [arch=vit_b16_224|clip_l14_336] [dtype=f16|bf16] [frames] [iters] [layers]"""
import sys, time, torch
sys.path.insert(0, ".")
import vidmem  # noqa
from _dev import maybe_dev; maybe_dev()   # TOOLS_DEV=1: the developer library (VIDMEM_* switches)
from vidmem import specs, synthetic as syn
from vidmem.encoder import FrameEncoder
arch = sys.argv[1] if len(sys.argv) > 1 else "vit_b16_224"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 441
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
spec = specs.SPECS[arch]
if len(sys.argv) > 5:
    spec = dict(spec, layers=int(sys.argv[5]))
enc = FrameEncoder(spec, syn.encoder_weights(spec, seed=42), dtype=dtype, device=0)
g = torch.Generator(device="cuda").manual_seed(1)
S = spec["image"]
fr = torch.randint(0, 256, (B, S, S, 3), generator=g, device="cuda", dtype=torch.uint8)
for _ in range(2):
    e = enc.embed_frames(fr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    e = enc.embed_frames(fr)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"{arch} {dtype}: {B} frames in {dt * 1e3:.2f} ms = {B / dt:.0f} frames/s, "
      f"{B / dt * specs.flops_per_frame(spec) / 1e12:.0f} TFLOP/s (algorithmic)")
enc.ctx.profile_enable(20000)
e = enc.embed_frames(fr)
bd = enc.ctx.profile_read()
print({k: (round(v[0], 3), v[1]) for k, v in bd.items() if v[1]})
