#!/usr/bin/env python3
"""The RCCL branch of dist.ShardedRetriever on ONE GPU: a one-rank ``nccl`` process group, both all-gathers forced.

What it proves before an 8-GPU node ever sees this code: librccl loads and initialises, the three dtypes of
the exchange (fp16 queries, fp64 scores, int64 rows) all-gather on device tensors, and the stream ordering around
``vm_topk_merge`` holds - the forced-collective search must equal the local search bit for bit.  It also times the two
all-gathers at the sizes a rank of the 8-GPU job brings (its own share: one rank's payload).

Runs as a process of its own (bench.py and tests/test_dist_gpu.py start it as a child with a timeout): the process group is
created before any other GPU work, nothing is re-executed, and a hang cannot take the caller with it.
Prints one JSON line; exit code 0 = results identical.  ``--oracle`` (tests only) also checks against oracle/cref.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000)
    ap.add_argument("--queries", type=int, default=880)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--oracle", action="store_true", help="also check against oracle/cref (tests only)")
    args = ap.parse_args()
    os.environ.setdefault("MASTER_PORT", str(free_port()))

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    warm = torch.ones(8, device=dev)
    dist.all_reduce(warm)                      # communicator creation happens on the first collective
    torch.cuda.synchronize()
    init_s = time.perf_counter() - t0

    import vidmem  # noqa: F401
    from vidmem.dist import ShardedRetriever
    from vidmem.memory import EmbeddingMemory
    D, M, Q, k = args.dim, args.rows, args.queries, args.topk
    g = torch.Generator(device=dev).manual_seed(5)
    rows = torch.randn((M, D), generator=g, device=dev)
    rows = (rows / rows.norm(dim=1, keepdim=True)).to(torch.float16)
    rows[M // 2 + 1] = rows[17]                       # an exact tie: (score descending, row ascending) must hold
    rows[1000:1040] = rows[999]                       # 40 identical rows: uncertifiable on the fast path -> redo
    q = torch.randn((Q, D), generator=g, device=dev).to(torch.float16)
    q[3] = rows[17]
    q[5] = rows[999]
    mem = EmbeddingMemory(M, D, "f16")
    mem.append(rows)
    forced = ShardedRetriever(mem, 0, 1, force_collectives=True)
    s_f, r_f = forced.search(q, k)
    s_l, r_l = mem.topk(q, k)
    same = bool(torch.equal(s_f, s_l) and torch.equal(r_f, r_l))
    ok = same and r_f[3, :2].tolist() == [17, M // 2 + 1] and r_f[5].tolist() == list(range(999, 999 + k))
    oracle_ok = None
    if args.oracle:
        import numpy as np
        from oracle import cref
        bits = lambda t: t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)   # noqa: E731
        nq = min(Q, 32)
        o_r, o_s = cref.cosine_topk(bits(q[:nq]), bits(rows), k, dtype="f16")
        oracle_ok = bool(np.array_equal(r_f[:nq].cpu().numpy(), o_r) and np.array_equal(s_f[:nq].cpu().numpy(), o_s))
        ok = ok and oracle_ok

    def timed(t):
        out = torch.empty_like(t)
        for _ in range(3):
            dist.all_gather_into_tensor(out, t)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.reps):
            dist.all_gather_into_tensor(out, t)
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / args.reps * 1e3

    ms_q = timed(q)
    ms_s = timed(s_l)
    ms_r = timed(r_l)
    t1 = time.perf_counter()
    for _ in range(args.reps):
        forced.search(q, k)
    torch.cuda.synchronize()
    ms_forced = (time.perf_counter() - t1) / args.reps * 1e3
    t1 = time.perf_counter()
    for _ in range(args.reps):
        mem.topk(q, k)
    torch.cuda.synchronize()
    ms_local = (time.perf_counter() - t1) / args.reps * 1e3
    print(json.dumps({
        "backend": dist.get_backend(), "world": 1, "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()),
        "init_seconds": init_s, "results_identical_to_local_search": same, "oracle_identical": oracle_ok, "ok": ok,
        "shape": {"rows": M, "queries": Q, "dim": D, "k": k},
        "all_gather_ms": {"queries_f16": ms_q, "scores_f64": ms_s, "rows_i64": ms_r},
        "all_gather_bytes": {"queries_f16": Q * D * 2, "scores_f64": Q * k * 8, "rows_i64": Q * k * 8},
        "search_ms": {"forced_collectives": ms_forced, "local": ms_local},
        "uncertified_queries_redone": mem.uncertified_count,
    }))
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
