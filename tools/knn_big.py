import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vidmem
from vidmem.memory import EmbeddingMemory
D, k, M = 768, 10, 100_000
mem = EmbeddingMemory(M, D, "f16")
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.randn((M, D), generator=g, device="cuda"); mem.append((x / x.norm(dim=1, keepdim=True)).to(torch.float16))
for Q in (880, 1760, 3520, 7040):
    q = torch.randn((Q, D), generator=g, device="cuda").to(torch.float16)
    for _ in range(3): mem.topk(q, k)
    torch.cuda.synchronize(); mem.ctx.profile_enable(2048)
    t0 = time.perf_counter()
    for _ in range(10): mem.topk(q, k)
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 10
    p = mem.ctx.profile_read(); mem.ctx.profile_enable(0)
    print(f"Q={Q}: launch {dtm*1e3:.3f} ms  " + "  ".join(f"{c} {v[0]/10:.3f}" for c, v in p.items() if v[0] > 0), f" uncert {mem.uncertified_count}", flush=True)
