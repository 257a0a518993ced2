"""Row-sharded retrieval across the GPUs of one node.

The reference is single-process (one asyncio loop, src/cli/main.py:239-284) and its memory lives in one Neo4j
instance; here frames are sharded by chunk and memory rows by owner rank, one process per GPU.  Global row id of
local row r on rank g is ``r * world + g`` so the (score desc, row id asc) tie rule stays global.

Exchange per search (the only collectives on the path, RCCL over xGMI; backend "nccl" is RCCL on ROCm):
  1. all-gather of the step's query embeddings  [F, D] -> [world*F, D]      (24 KB per 16-frame chunk)
  2. local cosine top-k of ALL queries over the local shard                 (csrc/topk.hip; queries the fp32 scan
     cannot certify are redone exhaustively on the device, csrc/topk_exact.hip, so every shard's list is exact)
  3. all-gather of the candidates {fp64 score, int64 global row} [world*F, k]
  4. merge of the `world` candidate lists of this rank's own queries        (csrc/topk.hip topk_merge_kernel)
No bulk row traffic ever crosses GPUs.  With world == 1 there is no collective and no torch.distributed import -
unless ``force_collectives`` asks for them: a one-rank group still runs both all-gathers through the backend (RCCL
loads, gathers fp16 / fp64 / int64 device tensors, and the stream ordering around ``vm_topk_merge`` is the N > 1 one),
which is how a one-GPU box exercises the ``nccl`` branch (tests/test_dist_gpu.py, tools/rccl_world1.py).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch


class ShardedRetriever:
    def __init__(self, memory, rank: int = 0, world: int = 1, group=None,
                 local_topk: Optional[Callable] = None, merge: Optional[Callable] = None,
                 force_collectives: bool = False):
        """``local_topk(queries, k, row_stride, row_offset) -> (scores, rows)`` and
        ``merge(scores[parts,Q,k], rows[parts,Q,k]) -> (scores[Q,k], rows[Q,k])`` default to the HIP kernels;
        the CPU gloo tests inject checkers to exercise the sharding logic without a GPU."""
        self.memory = memory
        self.rank, self.world, self.group = int(rank), int(world), group
        self.force_collectives = bool(force_collectives)
        if local_topk is None:
            def local_topk(q, k, stride, offset):
                # exhaustive answer on the local shard: the scan's uncertified queries are redone on the device
                # (vm_topk_redo_flagged) BEFORE the candidate all-gather; no host read-back, no cross-rank branch
                return memory.topk(q, k, row_stride=stride, row_offset=offset)
        if merge is None:
            from .memory import topk_merge

            def merge(s, r):
                return topk_merge(memory.ctx, s, r)
        self._local_topk, self._merge = local_topk, merge

    def search(self, queries: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """queries [F, D] (this rank's) -> global top-k (scores [F,k] fp64, global rows [F,k] int64)."""
        if self.world == 1 and not self.force_collectives:
            return self._local_topk(queries, k, 1, 0)
        F = queries.shape[0]
        q_all = self._all_gather(queries)                                   # [world*F, D], rank-major
        s_loc, r_loc = self._local_topk(q_all, k, self.world, self.rank)
        QA = s_loc.shape[0]
        s_all = self._all_gather(s_loc).view(self.world, QA, k)
        r_all = self._all_gather(r_loc).view(self.world, QA, k)
        lo = self.rank * F
        return self._merge(s_all[:, lo:lo + F].contiguous(), r_all[:, lo:lo + F].contiguous())

    def _all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """Rank-major concatenation along dim 0.  RCCL gathers device tensors in place; under a gloo group (CPU tests,
        single-GPU rehearsal of the N>1 path) device tensors are staged through the host, since gloo has no device
        all-gather."""
        import torch.distributed as dist
        t = t.contiguous()
        out_shape = (self.world * t.shape[0],) + tuple(t.shape[1:])
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            host = torch.empty(out_shape, dtype=t.dtype)
            dist.all_gather_into_tensor(host, t.cpu(), group=self.group)
            return host.to(t.device)
        out = torch.empty(out_shape, dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        return out

    def uncertified_total(self) -> int:
        """Local-shard queries the fast scan could not certify and the exhaustive kernel therefore redid (since the
        memory's ``reset_uncertified``).  Informational: the returned results are exhaustive either way."""
        return int(getattr(self.memory, "uncertified_count", 0))
