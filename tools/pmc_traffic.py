#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc runs of bench.py (FETCH_SIZE and WRITE_SIZE, separate passes as
MI355X_MICROARCH.md prescribes: both do not fit one pass) into profiles/<name>.json: per-kernel average HBM/fabric
bytes per launch.  gfx950 correction: FETCH_SIZE tallies 64 B per 128-B read request -> doubled; WRITE_SIZE is exact.
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [leg=shape ...]
  leg=shape: the launch shapes the profiled bench run used, e.g. main=F880,mb441,R100000,k10 c3=F2128,mb112 - bench.py
  only quotes a kernel's traffic when its own run has the same shape (bench.py pmc_traffic)."""
import collections, csv, json, sys

def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "at::native" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

def short(name):
    """Strip the trailing argument list (balanced parentheses from the end)."""
    if not name.endswith(")"):
        return name
    depth = 0
    for i in range(len(name) - 1, -1, -1):
        depth += name[i] == ")"
        depth -= name[i] == "("
        if depth == 0:
            return name[:i].strip()
    return name

fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"unit": "bytes per launch (average)", "correction": "read = 2 * FETCH_SIZE * 1024 (gfx950), write = WRITE_SIZE * 1024",
       "shapes": dict(a.split("=", 1) for a in sys.argv[4:]), "kernels": {}}
for k in sorted(fetch):
    f, n = fetch[k]
    w = write.get(k, (0.0, 0))[0]
    out["kernels"][short(k)] = {"launches": n, "read_bytes": 2 * f * 1024, "write_bytes": w * 1024,
                                                "traffic_bytes": 2 * f * 1024 + w * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k[:70]:70s} n={v['launches']:4d} read {v['read_bytes']/1e6:9.1f} MB  write {v['write_bytes']/1e6:9.1f} MB")
