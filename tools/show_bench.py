"""Print the headline numbers of a bench.py JSON line (helper for gpurun one-liners)."""
import json, sys
for path in sys.argv[1:]:
    for line in open(path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        r = d.get("roofline") or {}
        print(path, "value", round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 3),
              "| roofline", r.get("kernel"), round(r.get("achieved", 0), 1), r.get("unit"),
              "frac", round(r.get("frac", 0), 3), "avg_ms", round(r.get("avg_launch_ms", 0), 4))
