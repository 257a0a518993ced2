#!/usr/bin/env python3
"""Fold a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES ...) of bench.py
into profiles/<name>.json: per kernel instantiation, the average per launch and the matrix-pipe utilisation
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs)
(MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, = 16 per v_mfma_f32_16x16x32 per SIMD; rocprofv3 reports
GRBM_GUI_ACTIVE summed over the 8 XCDs).  usage: pmc_mfma.py <counter_collection.csv> <out.json> [note]"""
import collections, csv, json, sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "at::native" in name or "rocclr" in name:
        continue
    agg[name.split("(")[0].strip() if not name.startswith("void (anonymous") else name[:name.rfind("(")].strip()][
        r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"unit": "average per launch", "simds": 1024, "xcds": 8, "note": sys.argv[3] if len(sys.argv) > 3 else "",
       "kernels": {}}
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    m["launches"] = len(next(iter(d.values())))
    if m.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        m["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8)
    out["kernels"][k] = m
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, m in out["kernels"].items():
    if m.get("mfma_util"):
        print(f"{k[:72]:72s} n={m['launches']:4d} mfma_util {m['mfma_util']:.3f}")
