"""Cuts a small excerpt out of the ONE graph export the reference ships (data/exports/mvp_93e9c82e-...json, written by
GraphExporter.export_graph, src/components/graph_exporter.py:42-79) into tests/golden/export_excerpt.json, so the
export / import bridge (vidmem.bridge) is pinned to a real instance of the format and not to its own output.

    python tests/golden/make_export_golden.py          (authoring container; /root/reference is not on the GPU box)

Kept: the top-level keys and values as they are, the first 4 Chunk nodes, the first 3 Entity nodes and the first 3
relationships with every key intact.  Long strings (captions, id lists) are cut to 160 characters / 4 items so the
fixture stays a few KB; nothing else is altered.  The file is DATA the reference's own run produced - none of its
source text.
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = "/root/reference/data/exports/mvp_93e9c82e-95d6-4864-8ac1-2ae70edfd961.json"


def trim(v):
    if isinstance(v, str) and len(v) > 160:
        return v[:160]
    if isinstance(v, list):
        return [trim(x) for x in v[:4]]
    if isinstance(v, dict):
        return {k: trim(x) for k, x in v.items()}
    return v


def main():
    d = json.load(open(SRC, encoding="utf-8"))
    chunks = [n for n in d["nodes"] if "Chunk" in n["labels"]][:4]
    ents = [n for n in d["nodes"] if "Entity" in n["labels"]][:3]
    out = {k: v for k, v in d.items() if k not in ("nodes", "relationships")}
    out["nodes"] = [trim(n) for n in chunks + ents]
    out["relationships"] = [trim(r) for r in d["relationships"][:3]]
    out = {k: out[k] for k in d}                       # original key order
    stats = {"source_nodes": len(d["nodes"]), "source_relationships": len(d["relationships"]),
             "source_chunk_nodes": sum("Chunk" in n["labels"] for n in d["nodes"]),
             "source_chunks_with_embedding": sum(bool(n["properties"].get("embedding")) for n in d["nodes"]
                                                 if "Chunk" in n["labels"]),
             "chunk_property_keys": sorted({k for n in d["nodes"] if "Chunk" in n["labels"] for k in n["properties"]})}
    path = os.path.join(ROOT, "tests", "golden", "export_excerpt.json")
    json.dump({"excerpt": out, "stats": stats}, open(path, "w", encoding="utf-8"), indent=1, ensure_ascii=False)
    print("wrote", path, os.path.getsize(path), "bytes;", stats)


if __name__ == "__main__":
    main()
