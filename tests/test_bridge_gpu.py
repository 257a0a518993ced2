"""GPU: the device memory survives a trip through every persistence shape bit for bit:
native snapshot/restore, the exporter's JSON, the _get_chunk_embeddings dict."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _filled(ring=False, cap=300, n=260, D=256, dtype="f16"):
    from vidmem.memory import EmbeddingMemory
    g = torch.Generator().manual_seed(9)
    rows = torch.randn((n, D), generator=g)
    mem = EmbeddingMemory(cap, D, dtype, ring=ring, graph_uuid="g-7")
    for lo in range(0, n, 64):
        hi = min(n, lo + 64)
        mem.append(rows[lo:hi], ids=[f"g-7_{lo // 64}_{i}" for i in range(hi - lo)],
                   meta=[{"content": f"c{r}", "time": "00:00-00:02"} for r in range(lo, hi)])
    q = torch.randn((6, D), generator=g)
    return mem, q


def _same_answers(a, b, q, offset=0):
    sa, ra = a.topk(q, 7)
    sb, rb = b.topk(q, 7)
    assert np.array_equal(sa.cpu().numpy(), sb.cpu().numpy())
    assert np.array_equal(ra.cpu().numpy() - offset, rb.cpu().numpy())
    assert [a.id_of(r) for r in ra[0].tolist()] == [b.id_of(r) for r in rb[0].tolist()]


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_snapshot_restore_bit_identical(tmp_path, dtype):
    from vidmem.memory import EmbeddingMemory
    mem, q = _filled(dtype=dtype)
    path = str(tmp_path / "mem.npz")
    mem.snapshot(path)
    back = EmbeddingMemory.restore(path)
    assert back.graph_uuid == "g-7" and len(back) == len(mem) and back.ids == mem.ids and back.meta == mem.meta
    assert np.array_equal(back.rows_host()[1], mem.rows_host()[1])
    _same_answers(mem, back, q)


def test_ring_snapshot_keeps_row_order_of_survivors(tmp_path):
    from vidmem.memory import EmbeddingMemory
    mem, q = _filled(ring=True, cap=100, n=260)
    base, rows = mem.rows_host()
    assert base == 160 and rows.shape[0] == 100
    path = str(tmp_path / "ring.npz")
    mem.snapshot(path)
    back = EmbeddingMemory.restore(path)          # row ids restart at 0: old row r -> r - 160
    assert back.ids == mem.ids[160:]
    _same_answers(mem, back, q, offset=160)


def test_export_json_and_chunk_dict_round_trip(tmp_path):
    from vidmem import bridge
    from vidmem.memory import EmbeddingMemory
    mem, q = _filled()
    path = bridge.write_export(mem, str(tmp_path / "export.json"), embedding_model="vit-b16")
    back = EmbeddingMemory(300, 256, "f16", graph_uuid="g-7")
    n, skipped = bridge.load_export(back, path)
    assert n == len(mem) and not skipped
    _same_answers(mem, back, q)
    # the dict _get_chunk_embeddings would return for this graph, rebuilt from what _create_chunks_... would store
    chunks = bridge.chunks_for_neo4j(mem)
    again = EmbeddingMemory(300, 256, "f16")
    n, skipped = bridge.load_chunk_embeddings(again, {c["id"]: c["embedding"] for c in chunks})
    assert n == len(mem) and not skipped
    _same_answers(mem, again, q)


def test_ring_host_tables_stay_bounded():
    """A rolling window keeps ids / meta for the resident rows only (plus slack): after many wraps the tables are a
    bounded suffix, id_of / meta_of still resolve every resident row, overwritten rows resolve to None, and the
    snapshot / Neo4j shapes are unaffected."""
    from vidmem import bridge
    from vidmem.memory import EmbeddingMemory
    cap, D, B = 64, 256, 48
    mem = EmbeddingMemory(cap, D, "f16", ring=True)
    g = torch.Generator().manual_seed(4)
    total = 0
    for step in range(40):                                   # 1920 rows through a 64-row ring
        rows = torch.randn((B, D), generator=g)
        first = mem.append(rows, ids=[f"g_{step}_{i}" for i in range(B)], meta=[{"content": f"{step}.{i}"} for i in range(B)])
        assert first == total
        total += B
    assert len(mem) == total and len(mem.ids) <= 2 * cap + 1024 + B and mem.table_base > 0
    lo = total - cap
    assert mem.id_of(total - 1) == "g_39_47" and mem.id_of(lo) == f"g_{lo // B}_{lo % B}"
    assert mem.meta_of(total - 1) == {"content": "39.47"} and mem.id_of(mem.table_base - 1) is None
    q = torch.randn((3, D), generator=g)
    s, r = mem.topk(q, 5)
    assert (r >= lo).all() and all(mem.id_of(int(x)) is not None for x in r.flatten())
    chunks = bridge.chunks_for_neo4j(mem)                    # resident rows only, ids intact
    assert len(chunks) == cap and chunks[-1]["id"] == "g_39_47" and chunks[0]["index"] == lo % B
    assert mem.sync() == total and mem.id_of(total - 1) == "g_39_47"
