import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"]), "fps", round(d["ms_per_step"], 2), "ms/step", d.get("kernel_time_ms_per_step"))
