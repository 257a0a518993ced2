"""Developer probe: which queries go to the exhaustive redo when the memory already holds exact copies of them?
Random unit vectors (far from everything but their copies) and encoder outputs of noise frames (all near each other)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vidmem
from vidmem import specs, synthetic as syn
from vidmem.encoder import FrameEncoder
from vidmem.memory import EmbeddingMemory
M, D, k = 100_000, 768, 10
g = torch.Generator(device="cuda").manual_seed(7)
spec = specs.VIT_B16_224
enc = FrameEncoder(spec, syn.encoder_weights(spec, seed=42), "f16")
def fresh():
    mem = EmbeddingMemory(M, D, "f16", ring=True)
    x = torch.randn((M, D), generator=g, device="cuda"); mem.append((x / x.norm(dim=1, keepdim=True)).to(torch.float16))
    return mem
for kind in ("random", "noise-frame embeddings"):
    mem = fresh()
    if kind == "random":
        q = torch.randn((880, D), generator=g, device="cuda"); q = (q / q.norm(dim=1, keepdim=True)).to(torch.float16)
        other = [torch.randn((880, D), generator=g, device="cuda") for _ in range(4)]
        other = [(o / o.norm(dim=1, keepdim=True)).to(torch.float16) for o in other]
    else:
        fr = torch.randint(0, 256, (5, 880, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)
        q = enc.embed_frames(fr[0]); other = [enc.embed_frames(fr[i]) for i in range(1, 5)]
        c = (q[:64].float() @ q[64:128].float().T)
        print(f"  cos between different noise frames: mean {c.mean():.4f} min {c.min():.4f} max {c.max():.4f}")
    for o in other: mem.append(o)           # thousands of neighbours of the same kind, none equal to a query
    for copies in range(0, 3):
        mem.reset_uncertified()
        s, r = mem.topk(q, k)
        flags = mem._scratch.flags[:880].clone()
        torch.cuda.synchronize()
        nf = int(flags.sum())
        print(f"{kind}: copies of the queries in memory: {copies}  redone {mem.uncertified_count} flagged {nf}  "
              f"scores of query 0: {[round(v, 7) for v in s[0, :6].tolist()]}", flush=True)
        if nf:
            i = int(flags.nonzero()[0])
            print(f"    first flagged query {i}: scores {[round(v, 9) for v in s[i].tolist()]} rows {r[i].tolist()}")
        mem.append(q)
    mem.close()
