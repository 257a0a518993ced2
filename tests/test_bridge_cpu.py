"""Host logic of the Neo4j / export bridge against a host stand-in for the device memory.

The shapes checked are the reference's: text_chunks dicts (src/components/neo4j_handler.py:217-253), the
_get_chunk_embeddings dict and its guards (src/components/pre_llm_injector.py:390-412), the exporter's node dicts
(src/components/graph_exporter.py:60-66,97-101)."""
import json

import numpy as np
import pytest
import torch

from vidmem import bridge


class HostMemory:
    """What bridge.py needs from EmbeddingMemory, on the host (same single rounding to the 16-bit dtype)."""

    def __init__(self, dim, dtype="f16", graph_uuid=None):
        self.dim, self.dtype_name, self.graph_uuid = dim, dtype, graph_uuid
        self.ids, self.meta = [], []
        self._rows = np.zeros((0, dim), np.uint16)

    def append(self, rows, ids=None, meta=None):
        t = torch.tensor(rows, dtype=torch.float32).to(torch.float16 if self.dtype_name == "f16" else torch.bfloat16)
        first = len(self.ids)
        self._rows = np.concatenate([self._rows, t.view(torch.int16).numpy().view(np.uint16)])
        self.ids += list(ids)
        self.meta += list(meta)
        return first

    def rows_host(self):
        return 0, self._rows

    def id_of(self, r):
        return self.ids[r] if 0 <= r < len(self.ids) else None

    def meta_of(self, r):
        return self.meta[r] if 0 <= r < len(self.meta) else None


def _vecs(n, d, seed=0):
    return np.random.default_rng(seed).standard_normal((n, d)).tolist()


def test_load_chunk_embeddings_guards_and_order():
    mem = HostMemory(8)
    v = _vecs(4, 8)
    got = {"run_0_1": v[0], "": v[1], "run_0_2": "not a list", "run_0_3": [], "run_0_4": v[2][:5], "run_1_0": v[3]}
    n, skipped = bridge.load_chunk_embeddings(mem, got)
    assert n == 2 and mem.ids == ["run_0_1", "run_1_0"]          # dict order, guards of :404-406 / :362 / :378
    assert skipped == ["", "run_0_2", "run_0_3", "run_0_4"]
    want = torch.tensor([v[0], v[3]], dtype=torch.float32).to(torch.float16).to(torch.float64).tolist()
    assert bridge.rows_to_lists(mem.rows_host()[1], "f16") == want


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_chunks_for_neo4j_shape_and_exact_values(dtype):
    mem = HostMemory(16, dtype)
    v = _vecs(3, 16, seed=1)
    mem.append(v, ids=["u_2_0", "u_2_1", "free-form"], meta=[{"content": "a", "time": "00:00-00:02"}, None, {}])
    chunks = bridge.chunks_for_neo4j(mem)
    assert [c["id"] for c in chunks] == ["u_2_0", "u_2_1", "free-form"]
    assert [c["index"] for c in chunks] == [0, 1, None]          # last "_" field of the id scheme, :91
    assert [c["content"] for c in chunks] == ["a", None, None]
    assert set(chunks[0]) == {"id", "content", "index", "embedding"}
    td = torch.float16 if dtype == "f16" else torch.bfloat16
    want = torch.tensor(v, dtype=torch.float32).to(td).to(torch.float64).tolist()
    assert [c["embedding"] for c in chunks] == want and all(isinstance(x, float) for x in chunks[0]["embedding"])
    assert bridge.chunks_for_neo4j(mem, 1, 1)[0]["id"] == "u_2_1"
    with pytest.raises(IndexError):
        bridge.chunks_for_neo4j(mem, 2, 5)


def test_export_round_trip_is_identity(tmp_path):
    mem = HostMemory(12, "f16", graph_uuid="g-1")
    mem.append(_vecs(5, 12, seed=2), ids=[f"g-1_0_{i}" for i in range(5)],
               meta=[{"content": f"c{i}", "time": "00:00-00:02"} for i in range(5)])
    path = bridge.write_export(mem, str(tmp_path / "export.json"), embedding_model="vit-b16", batch_id=3)
    data = json.load(open(path))
    assert data["export_format_version"] == "1.0" and data["graph_uuid"] == "g-1" and data["relationships"] == []
    node = data["nodes"][0]
    assert node["labels"] == ["Chunk"] and "graph_uuid" not in node["properties"]
    assert node["properties"]["embedding_model"] == "vit-b16" and node["properties"]["batch_id"] == 3
    # foreign nodes and an embedding-less chunk (neo4j_handler.py:243-253) are ignored on the way back in
    data["nodes"].insert(1, {"name": "Bob", "labels": ["Entity"], "properties": {"id": "e1"}})
    data["nodes"].append({"name": None, "labels": ["Chunk"], "properties": {"id": "g-1_1_0", "content": "x"}})
    back = HostMemory(12, "f16", graph_uuid="g-1")
    n, skipped = bridge.load_export(back, data)
    assert n == 5 and skipped == [] and back.ids == mem.ids
    assert np.array_equal(back.rows_host()[1], mem.rows_host()[1])  # bit-identical rows
    assert back.meta[2] == {"content": "c2", "time": "00:00-00:02", "batch_id": 3, "created_at": None}
    assert bridge.export_nodes(back, embedding_model="vit-b16", batch_id=3) == json.load(open(path))["nodes"]


def test_load_export_rejects_wrong_graph_and_version():
    mem = HostMemory(4, graph_uuid="g-1")
    with pytest.raises(ValueError):
        bridge.load_export(mem, {"graph_uuid": "g-2", "nodes": [], "export_format_version": "1.0"})
    with pytest.raises(ValueError):
        bridge.load_export(mem, {"graph_uuid": "g-1", "nodes": [], "export_format_version": "2.0"})
    n, skipped = bridge.load_export(mem, {"graph_uuid": "g-1", "export_format_version": "1.0", "nodes": [
        {"labels": ["Chunk"], "properties": {"id": "a", "embedding": [1.0, 2.0]}}]})
    assert n == 0 and skipped == ["a"]                              # wrong dimension: reported, not appended


def test_bf16_rows_to_lists_matches_torch():
    t = torch.randn(7, 9, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16)
    raw = t.view(torch.int16).numpy().view(np.uint16)
    assert bridge.rows_to_lists(raw, "bf16") == t.to(torch.float64).tolist()


def test_export_format_is_pinned_to_the_reference_artifact(tmp_path):
    """tests/golden/export_excerpt.json is cut (tests/golden/make_export_golden.py) out of the ONE export the
    reference ships, data/exports/mvp_93e9c82e-...json: 326 Chunk nodes, none with an embedding (the MVP run stored
    its chunks through the no-embedding branch, src/components/neo4j_handler.py:243-253).
      * load_export must accept it: nothing embedded -> nothing appended, nothing skipped, nothing raised;
      * write_export must emit the same top-level keys in the same order (src/components/graph_exporter.py:61-67,
        export_timestamp included) and Chunk nodes of the same shape (:97-101) whose property keys cover the
        reference's (batch_id, created_at, id, content)."""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    fx = json.load(open(os.path.join(here, "golden", "export_excerpt.json"), encoding="utf-8"))
    ref, stats = fx["excerpt"], fx["stats"]
    assert stats["source_chunk_nodes"] == 326 and stats["source_chunks_with_embedding"] == 0
    assert list(ref) == ["graph_uuid", "export_timestamp", "nodes", "relationships", "export_format_version"]
    assert ref["export_format_version"] == bridge.EXPORT_FORMAT_VERSION

    mem = HostMemory(8, graph_uuid=ref["graph_uuid"])
    n, skipped = bridge.load_export(mem, ref)
    assert (n, skipped, mem.ids) == (0, [], [])
    with pytest.raises(ValueError):
        bridge.load_export(HostMemory(8, graph_uuid="another-graph"), ref)

    ref_chunk = next(x for x in ref["nodes"] if "Chunk" in x["labels"])
    assert set(ref_chunk) == {"name", "labels", "properties"} and ref_chunk["name"] is None
    assert set(ref_chunk["properties"]) == set(stats["chunk_property_keys"]) == {"batch_id", "created_at", "id", "content"}

    v = _vecs(2, 8, seed=4)
    mem.append(v, ids=[ref_chunk["properties"]["id"], "x_0_1"],
               meta=[{"content": ref_chunk["properties"]["content"], "batch_id": ref_chunk["properties"]["batch_id"],
                      "created_at": ref_chunk["properties"]["created_at"]}, {"content": "b"}])
    out = json.load(open(bridge.write_export(mem, str(tmp_path / "out.json")), encoding="utf-8"))
    assert list(out) == list(ref)                                     # same keys, same order
    assert isinstance(out["export_timestamp"], str) and len(out["export_timestamp"]) == len(ref["export_timestamp"])
    node = out["nodes"][0]
    assert set(node) == set(ref_chunk) and node["labels"] == ref_chunk["labels"] and node["name"] is None
    assert set(ref_chunk["properties"]) <= set(node["properties"])     # + embedding, which this path adds
    for k in ("batch_id", "created_at", "id", "content"):
        assert node["properties"][k] == ref_chunk["properties"][k]
    # and the file round-trips through the importer: rows and metadata identical
    back = HostMemory(8, graph_uuid=ref["graph_uuid"])
    assert bridge.load_export(back, out)[0] == 2
    assert np.array_equal(back.rows_host()[1], mem.rows_host()[1]) and back.ids == mem.ids
    assert back.meta[0]["batch_id"] == ref_chunk["properties"]["batch_id"]
    assert back.meta[0]["created_at"] == ref_chunk["properties"]["created_at"]
