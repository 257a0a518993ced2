"""Developer probe: kNN launch time (scan / finalize split).  knn_probe.py [D=768] [dtype=f16|bf16] [k=10] [M=1000000]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vidmem
from _dev import maybe_dev; maybe_dev()
from vidmem.memory import EmbeddingMemory
D = int(sys.argv[1]) if len(sys.argv) > 1 else 768
dt = sys.argv[2] if len(sys.argv) > 2 else "f16"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
M = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
td = torch.float16 if dt == "f16" else torch.bfloat16
mem = EmbeddingMemory(M, D, dt)
g = torch.Generator(device="cuda").manual_seed(7)
for lo in range(0, M, 250_000):
    n = min(250_000, M - lo)
    x = torch.randn((n, D), generator=g, device="cuda"); mem.append((x / x.norm(dim=1, keepdim=True)).to(td))
QS = [int(x) for x in sys.argv[5].split(',')] if len(sys.argv) > 5 else [16, 64, 256]
for Q in QS:
    q = torch.randn((Q, D), generator=g, device="cuda").to(td)
    for _ in range(3): mem.topk(q, k)
    torch.cuda.synchronize(); mem.ctx.profile_enable(512)
    t0 = time.perf_counter()
    for _ in range(50): mem.topk(q, k)
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 50
    p = mem.ctx.profile_read(); mem.ctx.profile_enable(0)
    sc, fi = p["topk_scan"][0] / 50, p["topk_finalize"][0] / 50
    print(f"D={D} {dt} k={k} Q={Q}: launch {dtm*1e3:.3f} ms, scan {sc:.3f} ms ({M*D*2/sc/1e6:.0f} GB/s), finalize {fi:.3f} ms, {Q/dtm:.0f} q/s, uncert {mem.uncertified_count}")
