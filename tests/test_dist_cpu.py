"""CPU, world_size 2 over gloo: the row-sharded retrieval exchange (vidmem.dist.ShardedRetriever) with the HIP
kernels replaced by oracle checkers, so the sharding / global-id / all-gather / merge logic of the N>1 path is
covered without a GPU.  (The kernels themselves are covered by tests/test_topk_gpu.py, incl. the 2-shard merge.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import similarity_ref as S


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, D, M, F, k, q_out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vidmem  # noqa: F401
    from vidmem.dist import ShardedRetriever
    rng = np.random.default_rng(0)
    mem = rng.standard_normal((M, D)).astype(np.float16)
    mem[7] = mem[3]          # an exact tie across two different shards (rows 3 and 7 with world 2)
    queries = rng.standard_normal((world * F, D)).astype(np.float16)
    queries[0] = mem[3]
    shard = mem[rank::world].astype(np.float64)   # row r of the shard has global id r*world + rank

    def local_topk(q, kk, stride, offset):
        rows, scores = S.batch_topk_np(q.numpy().astype(np.float64), shard, kk)
        rows = np.where(rows >= 0, rows * stride + offset, -1)
        return torch.from_numpy(scores), torch.from_numpy(rows)

    def merge(s, r):  # [parts,Q,k] -> [Q,k] by (score desc, row asc); checker for vm_topk_merge
        parts, Q, kk = s.shape
        out_s = torch.zeros((Q, kk), dtype=torch.float64)
        out_r = torch.full((Q, kk), -1, dtype=torch.int64)
        for q in range(Q):
            cand = [(float(s[p, q, j]), int(r[p, q, j])) for p in range(parts) for j in range(kk) if r[p, q, j] >= 0]
            cand.sort(key=lambda t: (-t[0], t[1]))
            for j, (sc, ro) in enumerate(cand[:kk]):
                out_s[q, j], out_r[q, j] = sc, ro
        return out_s, out_r

    ret = ShardedRetriever(memory=None, rank=rank, world=world, local_topk=local_topk, merge=merge)
    mine = torch.from_numpy(queries[rank * F:(rank + 1) * F])
    scores, rows = ret.search(mine, k)
    want_rows, want_scores = S.batch_topk_np(queries[rank * F:(rank + 1) * F].astype(np.float64),
                                             mem.astype(np.float64), k)
    ok = np.array_equal(rows.numpy(), want_rows) and np.array_equal(scores.numpy(), want_scores)
    q_out.put((rank, bool(ok), rows[0].tolist() if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_retrieval_world2_gloo():
    world, D, M, F, k = 2, 64, 101, 3, 5   # odd M: ragged shards (51 / 50 rows)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, D, M, F, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results), results
    first = [r for rk, _, r in results if rk == 0][0]
    assert first[:2] == [3, 7]  # the cross-shard tie resolves to the lower GLOBAL row id


def test_single_rank_needs_no_process_group():
    import vidmem  # noqa: F401
    from vidmem.dist import ShardedRetriever
    calls = []

    def local_topk(q, k, stride, offset):
        calls.append((stride, offset))
        return torch.zeros((q.shape[0], k), dtype=torch.float64), torch.zeros((q.shape[0], k), dtype=torch.int64)

    r = ShardedRetriever(None, 0, 1, local_topk=local_topk, merge=lambda s, r: (s[0], r[0]))
    s, rows = r.search(torch.zeros((4, 8)), 3)
    assert calls == [(1, 0)] and s.shape == (4, 3)
