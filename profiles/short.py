"""Print the headline fields of bench.py JSON lines read from stdin (keeps gpurun tails readable)."""
import json
import sys

for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline", {})
    print(json.dumps({"value": d["value"], "ms_per_step": d["ms_per_step"], "n_gpus": d["n_gpus"],
                      "roofline": {k: r.get(k) for k in ("kernel", "achieved", "frac", "avg_us")},
                      "t2h_ms": d.get("t2h_kernels_ms_per_step"), "cpu": d.get("cpu_baseline", {}).get("value")}))
