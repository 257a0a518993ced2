"""Developer probe: the pieces of the extractor's one-search-per-group path (memory top-k, exact all-pairs, sort, mask)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vidmem
from vidmem.memory import EmbeddingMemory, topk_merge
D, F, k = 768, 880, 10
mem = EmbeddingMemory(120_000, D, "f16")
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((100_000, D), generator=g, device="cuda"); mem.append((x / x.norm(dim=1, keepdim=True)).to(torch.float16))
e = torch.randn((F, D), generator=g, device="cuda"); e = (e / e.norm(dim=1, keepdim=True)).to(torch.float16)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("topk", t(lambda: mem.topk(e, k)))
print("cosine_exact", t(lambda: mem.cosine_exact(e, e)))
gm = mem.cosine_exact(e, e)
print("sort stable", t(lambda: torch.sort(gm, dim=1, descending=True, stable=True)))
print("topk torch", t(lambda: torch.topk(gm, k, dim=1)))
starts = torch.repeat_interleave(torch.arange(55, device="cuda") * 16, 16)
print("mask", t(lambda: gm.masked_fill(torch.arange(F, device="cuda")[None, :] >= starts[:, None], float("-inf"))))
