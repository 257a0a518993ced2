"""Developer probe: kNN launch time (scan / finalize split) on a 1M x 768 index."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vidmem
from vidmem.memory import EmbeddingMemory
M, D = 1_000_000, 768
mem = EmbeddingMemory(M, D, "f16")
g = torch.Generator(device="cuda").manual_seed(7)
for lo in range(0, M, 250_000):
    x = torch.randn((250_000, D), generator=g, device="cuda"); mem.append((x / x.norm(dim=1, keepdim=True)).half())
for Q in (16, 64, 256):
    q = torch.randn((Q, D), generator=g, device="cuda").half()
    for _ in range(3): mem.topk(q, 10, check_certified=False)
    torch.cuda.synchronize(); mem.ctx.profile_enable(512)
    t0 = time.perf_counter()
    for _ in range(50): mem.topk(q, 10, check_certified=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    p = mem.ctx.profile_read(); mem.ctx.profile_enable(0)
    sc, fi = p["topk_scan"][0] / 50, p["topk_finalize"][0] / 50
    print(f"Q={Q}: launch {dt*1e3:.3f} ms, scan {sc:.3f} ms ({M*D*2/sc/1e6:.0f} GB/s), finalize {fi:.3f} ms, {Q/dt:.0f} q/s, uncert {int(mem._uncert.item())}")
