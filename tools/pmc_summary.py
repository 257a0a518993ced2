#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel (developer helper)."""
import collections, csv, sys
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        name = r['Kernel_Name']
        if 'at::native' in name or 'rocclr' in name: continue
        agg[name[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in sorted(agg.items()):
        n = len(next(iter(d.values())))
        print(f"{k}  (n={n})")
        print("    " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
