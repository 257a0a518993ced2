"""Developer probe for rocprofv3 --pmc: a few Q=64 / Q=256 top-k launches over a 1M x 768 index."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vidmem
from vidmem.memory import EmbeddingMemory
M, D = 1_000_000, 768
mem = EmbeddingMemory(M, D, "f16")
g = torch.Generator(device="cuda").manual_seed(7)
for lo in range(0, M, 250_000):
    x = torch.randn((250_000, D), generator=g, device="cuda"); mem.append((x / x.norm(dim=1, keepdim=True)).half())
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 64
q = torch.randn((Q, D), generator=g, device="cuda").half()
for _ in range(4): mem.topk(q, 10)
torch.cuda.synchronize()
print("done")
