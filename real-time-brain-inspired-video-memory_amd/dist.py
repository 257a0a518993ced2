"""Row-sharded retrieval across the GPUs of one node.

The reference is single-process (one asyncio loop, src/cli/main.py:239-284) and its memory lives in one Neo4j
instance; here frames are sharded by chunk and memory rows by owner rank, one process per GPU.  Global row id of
local row r on rank g is ``r * world + g`` so the (score desc, row id asc) tie rule stays global.

Exchange per search (the only collectives on the path, RCCL over xGMI; backend "nccl" is RCCL on ROCm):
  1. all-gather of the step's query embeddings  [F, D] -> [world*F, D]      (24 KB per 16-frame chunk)
  2. local cosine top-k of ALL queries over the local shard                 (csrc/topk.hip)
  3. all-gather of the candidates {fp64 score, int64 global row} [world*F, k]
  4. merge of the `world` candidate lists of this rank's own queries        (csrc/topk.hip topk_merge_kernel)
No bulk row traffic ever crosses GPUs.  With world == 1 there is no collective and no torch.distributed import.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch


class ShardedRetriever:
    def __init__(self, memory, rank: int = 0, world: int = 1, group=None,
                 local_topk: Optional[Callable] = None, merge: Optional[Callable] = None):
        """``local_topk(queries, k, row_stride, row_offset) -> (scores, rows)`` and
        ``merge(scores[parts,Q,k], rows[parts,Q,k]) -> (scores[Q,k], rows[Q,k])`` default to the HIP kernels;
        the CPU gloo tests inject checkers to exercise the sharding logic without a GPU."""
        self.memory = memory
        self.rank, self.world, self.group = int(rank), int(world), group
        if local_topk is None:
            def local_topk(q, k, stride, offset):
                return memory.topk(q, k, row_stride=stride, row_offset=offset, check_certified=False)
        if merge is None:
            from .memory import topk_merge

            def merge(s, r):
                return topk_merge(memory.ctx, s, r)
        self._local_topk, self._merge = local_topk, merge

    def search(self, queries: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """queries [F, D] (this rank's) -> global top-k (scores [F,k] fp64, global rows [F,k] int64)."""
        if self.world == 1:
            return self._local_topk(queries, k, 1, 0)
        import torch.distributed as dist
        F = queries.shape[0]
        q_all = torch.empty((self.world * F,) + tuple(queries.shape[1:]), dtype=queries.dtype,
                            device=queries.device)
        dist.all_gather_into_tensor(q_all, queries.contiguous(), group=self.group)
        s_loc, r_loc = self._local_topk(q_all, k, self.world, self.rank)
        QA = s_loc.shape[0]
        s_cat = torch.empty((self.world * QA, k), dtype=s_loc.dtype, device=s_loc.device)  # rank-major concatenation
        r_cat = torch.empty((self.world * QA, k), dtype=r_loc.dtype, device=r_loc.device)
        dist.all_gather_into_tensor(s_cat, s_loc.contiguous(), group=self.group)
        dist.all_gather_into_tensor(r_cat, r_loc.contiguous(), group=self.group)
        s_all, r_all = s_cat.view(self.world, QA, k), r_cat.view(self.world, QA, k)
        lo = self.rank * F
        return self._merge(s_all[:, lo:lo + F].contiguous(), r_all[:, lo:lo + F].contiguous())

    def uncertified_total(self) -> int:
        """Queries (since the last certified call) whose fast-path answer could not be proven exhaustive."""
        u = getattr(self.memory, "_uncert", None)
        return int(u.item()) if u is not None else 0
