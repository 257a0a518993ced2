"""Developer probe (GPU box): BASELINE configs[3]'s per-rank step with the search of step i on a SECOND stream beside
the encoder pass of step i+1 (same results: search i still sees exactly the rows of steps < i, append i follows it).
usage: c4_overlap_probe.py [steps=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vidmem  # noqa: F401
from vidmem import specs, synthetic as syn
from vidmem.encoder import FrameEncoder
from vidmem.memory import EmbeddingMemory, topk_merge

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
spec = specs.VIT_B16_224
D, k, F, W4, R4 = spec["hidden"], 10, 880, 8, 1_048_576
dev = torch.device("cuda", 0)
enc = FrameEncoder(spec, syn.encoder_weights(spec, seed=42), dtype="f16", device=0)
ctx = enc.ctx
g4 = torch.Generator(device=dev).manual_seed(4004)
shard = EmbeddingMemory(R4, D, "f16", ring=True, device=0)
for lo in range(0, R4, 262_144):
    x = torch.randn((262_144, D), generator=g4, device=dev)
    shard.append((x / x.norm(dim=1, keepdim=True)).to(torch.float16))
others = torch.cat([enc.embed_frames(torch.randint(0, 256, (F, 224, 224, 3), generator=g4, device=dev, dtype=torch.uint8))
                    for _ in range(W4 - 1)])
fr = torch.randint(0, 256, (4 * (n + 2), F, 224, 224, 3), generator=g4, device=dev, dtype=torch.uint8)


def search_append(emb):
    q_all = torch.cat([emb, others])
    s_l, r_l = shard.topk(q_all, k, row_stride=W4, row_offset=0)
    s_m, r_m = topk_merge(ctx, s_l.view(W4, F, k), r_l.view(W4, F, k))
    shard.append(emb)
    return s_m, r_m


def run(overlap, base):
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    res = []
    torch.cuda.synchronize()
    t0 = None
    for i in range(n + 2):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        emb = enc.embed_frames(fr[base + i])      # distinct frames in every run: no planted duplicates
        if overlap:
            ev = torch.cuda.Event()
            ev.record(main)
            emb.record_stream(side)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                res.append(search_append(emb))
        else:
            res.append(search_append(emb))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, res


for rep in range(2):
    a, ra = run(False, (2 * rep) * (n + 2))
    b, rb = run(True, (2 * rep + 1) * (n + 2))
    print(f"one stream {a:.2f} ms per step, search on a second stream {b:.2f} ms per step ({100 * (a / b - 1):+.1f} %)", flush=True)
print("uncertified redone:", shard.uncertified_count)
