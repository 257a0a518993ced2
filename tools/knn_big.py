"""Developer probe: many-query top-k (the per-shard load of an N-GPU step).  knn_big.py [M=1048576] [D=768] [k=10]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vidmem
from vidmem.memory import EmbeddingMemory
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_048_576
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
QS = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [256, 880, 1760, 3520, 7040]
mem = EmbeddingMemory(M, D, "f16")
g = torch.Generator(device="cuda").manual_seed(7)
for lo in range(0, M, 262_144):
    n = min(262_144, M - lo)
    x = torch.randn((n, D), generator=g, device="cuda"); mem.append((x / x.norm(dim=1, keepdim=True)).to(torch.float16))
for Q in QS:
    q = torch.randn((Q, D), generator=g, device="cuda").to(torch.float16)
    for _ in range(2): mem.topk(q, k)
    torch.cuda.synchronize(); mem.ctx.profile_enable(2048); mem.reset_uncertified()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n): mem.topk(q, k)
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / n
    p = mem.ctx.profile_read(); mem.ctx.profile_enable(0)
    sc = p["topk_scan"][0] / n
    print(f"M={M} D={D} Q={Q}: launch {dtm*1e3:.3f} ms  scan {sc:.3f} ms = {2.0*Q*M*D/sc/1e9:.0f} TFLOP/s ({2.0*Q*M*D/sc/1e9/2500:.3f} of 2.5 PF)  "
          + "  ".join(f"{c} {v[0]/n:.3f}" for c, v in p.items() if v[0] > 0), f" redone {mem.uncertified_count // n}", flush=True)
