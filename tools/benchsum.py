"""Developer helper: one-line summary of a bench.py JSON line (value, ms per step, roofline fraction, per-kernel ms).
usage: benchsum.py <bench.json>"""
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"]), "fps", round(d["ms_per_step"],2), "ms  frac", round(d["roofline"]["frac"],3))
k=d["kernel_time_ms_per_step"]; print({a:round(b,2) for a,b in k.items() if b>0.2})
