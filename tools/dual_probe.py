"""Developer probe: vm_encode's schedules (VM_ENC_OPT_SCHEDULE) against each other: embeddings bit for bit, frames/s
alternating A/B.  dual_probe.py [frames=1760] [arch] [dtype] [one_stream,two_streams] [micro_batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vidmem
from vidmem import specs, synthetic as syn
from vidmem.encoder import FrameEncoder
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1760
arch = sys.argv[2] if len(sys.argv) > 2 else "vit_b16_224"
dtype = sys.argv[3] if len(sys.argv) > 3 else "f16"
spec = specs.SPECS[arch]
w = syn.encoder_weights(spec, seed=42)
encs = {}
MODES = sys.argv[4].split(",") if len(sys.argv) > 4 else ["one_stream", "two_streams"]
MB = int(sys.argv[5]) if len(sys.argv) > 5 else 0
for mode in MODES:
    encs[mode] = FrameEncoder(spec, w, dtype, schedule=mode, micro_batch=MB)
S = spec["image"]
fr = torch.randint(0, 256, (F, S, S, 3), device="cuda", dtype=torch.uint8)
out = {m: e.embed_frames(fr) for m, e in encs.items()}
torch.cuda.synchronize()
print("bit-identical:", all(torch.equal(out[MODES[0]], o) for o in out.values()))
for rep in range(3):
    for m in MODES:
        e = encs[m]
        for _ in range(2): e.embed_frames(fr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6): e.embed_frames(fr)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 6
        print(f"{m} mb={MB}: {F} frames in {dt*1e3:.2f} ms = {F/dt:.0f} frames/s", flush=True)
