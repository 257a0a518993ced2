"""Developer probe: do exact duplicates of the queries among the stored rows send queries to the exhaustive redo?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vidmem
from vidmem.memory import EmbeddingMemory
M, D, k = 1_048_576, 768, 10
mem = EmbeddingMemory(M, D, "f16", ring=True)
g = torch.Generator(device="cuda").manual_seed(7)
for lo in range(0, M, 262_144):
    x = torch.randn((262_144, D), generator=g, device="cuda"); mem.append((x / x.norm(dim=1, keepdim=True)).to(torch.float16))
q = torch.randn((7040, D), generator=g, device="cuda")
q = (q / q.norm(dim=1, keepdim=True)).to(torch.float16)
for copies in range(0, 4):
    mem.reset_uncertified()
    s, r = mem.topk(q, k)
    torch.cuda.synchronize()
    print(f"copies of q[:880] in memory: {copies}  redone {mem.uncertified_count}  top scores {s[0, :4].tolist()}", flush=True)
    mem.append(q[:880])
