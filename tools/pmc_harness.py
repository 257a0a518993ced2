#!/usr/bin/env python3
"""Fold a `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace` pass over tools/bin/gemm_bench
into per-kernel-instantiation medians: duration, cycles per XCD, effective clock, matrix-pipe utilisation.
usage: pmc_harness.py <counter_collection.csv>"""
import collections, csv, statistics, sys
d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "gemm" not in n or "ref_gemm" in n:
        continue
    e = d.setdefault(r["Dispatch_Id"], {"name": n[n.find("gemm"):n.find("(GemmArgs")],
                                         "dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
g = collections.OrderedDict()
for e in d.values():
    g.setdefault(e["name"], []).append(e)
for n, es in g.items():
    cyc = statistics.median(x["GRBM_GUI_ACTIVE"] / 8 for x in es)
    dur = statistics.median(x["dur"] for x in es)
    mf = statistics.median(x["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (x["GRBM_GUI_ACTIVE"] / 8) for x in es)
    print(f"{n:36s} n={len(es):2d} dur {dur:7.1f} us  cycles {cyc:8.0f}  clock {cyc / dur / 1e3:.2f} GHz  mfma_util {mf:.3f}")
