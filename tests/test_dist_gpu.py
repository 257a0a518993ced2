"""GPU: the N>1 retrieval path end to end (HIP local top-k with row_stride/row_offset, all-gathers, vm_topk_merge) on two
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


def test_two_ranks_row_sharded_search_equals_single_memory():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "tests", "dist_worker.py")]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
