"""GPU: the RCCL branch on one rank with forced collectives, and the N>1 retrieval path end to end (HIP local top-k with row_stride/row_offset, all-gathers, vm_topk_merge) on two
ranks that share GPU 0 through a gloo group: global top-k of every rank's queries must equal, bit for bit, the top-k
over the unsharded memory.  (The driver's multi-GPU bench runs the same code over RCCL with one device per rank.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_rccl_branch_with_one_rank_and_forced_collectives():
    """backend "nccl" (= RCCL) with world_size 1: ShardedRetriever(force_collectives=True) sends fp16 queries, fp64
    scores and int64 rows through RCCL all-gathers on device tensors; the result must equal the local search and the C
    oracle bit for bit (exact ties and an uncertifiable tie flood included).  A process of its own: the group is created
    before any other GPU work and a hang is a timeout here, not a hung test session."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_world1.py"), "--oracle", "--rows", "60000",
                          "--queries", "96", "--reps", "5"], cwd=root, env=env, capture_output=True, text=True,
                         timeout=420)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    rep = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rep["backend"] == "nccl" and rep["results_identical_to_local_search"] and rep["oracle_identical"]
    assert rep["uncertified_queries_redone"] >= 1          # the tie flood went through the device-side redo


def test_two_ranks_row_sharded_search_equals_single_memory():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "tests", "dist_worker.py")]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
