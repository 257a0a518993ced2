"""Worker of tests/test_dist_gpu.py: one rank of a world-2 gloo group, both ranks on GPU 0 (rehearsal of the N>1 path:
RCCL needs one device per rank, which a 1-GPU test box does not have).  Exit code 0 = every check passed."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vidmem  # noqa: E402,F401
from vidmem.dist import ShardedRetriever  # noqa: E402
from vidmem.memory import EmbeddingMemory  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D, M, F, k = 256, 6000, 24, 7
    g = torch.Generator().manual_seed(5)                       # same data on every rank
    rows = torch.randn((M, D), generator=g).to(torch.float16)
    rows[4001] = rows[17]                                      # an exact duplicate on the OTHER shard: global tie rule
    queries = torch.randn((world * F, D), generator=g).to(torch.float16)
    queries[3] = rows[17]
    queries[F + 2] = rows[4001]
    # 40 identical rows, 20 per shard (tests/test_topk_gpu.py's uncertifiable case split across two shards): each
    # shard's scan must refuse to certify, redo exhaustively BEFORE the candidate all-gather, and the merged answer
    # must be the lowest ten row ids, alternating shards
    rows[1000:1040] = rows[999]
    queries[5] = rows[999]
    queries[F + 7] = rows[999]

    shard = EmbeddingMemory(M // world + 8, D, "f16")
    shard.append(rows[rank::world])                            # row r lives on rank r % world, local row r // world
    full = EmbeddingMemory(M, D, "f16")
    full.append(rows)

    mine = queries[rank * F:(rank + 1) * F].cuda()
    s, r = ShardedRetriever(shard, rank, world).search(mine, k)
    want_s, want_r = full.topk(mine, k)
    ok = torch.equal(r, want_r) and torch.equal(s, want_s)
    # ... and against the ORACLE (the unsharded HIP result above is itself only a GPU path)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import cref
    bits = lambda t: t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)
    o_r, o_s = cref.cosine_topk(bits(mine), bits(rows), k, dtype="f16")
    ok = ok and np.array_equal(r.cpu().numpy(), o_r) and np.array_equal(s.cpu().numpy(), o_s)
    tie_q = 5 if rank == 0 else 7
    ok = ok and r[tie_q].tolist() == list(range(999, 999 + k)) and shard.uncertified_count >= 1
    # the planted duplicates: rows 17 and 4001 tie exactly, the lower global row id comes first
    if rank == 0:
        ok = ok and r[3, :2].tolist() == [17, 4001]
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    if not ok:
        print(f"rank {rank}: sharded result differs from the single-memory result", file=sys.stderr)
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
