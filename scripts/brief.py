"""stdin: bench.py output; prints the few numbers compared during kernel tuning."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline", {})
    h = r.get("hbm_bound") or {}
    print("%-14s ms/step %.4f  triples/s %.0f  dense launch %.1f us  rows layer %.1f us  tiles %s  hbm-bound launch %.1f us" % (
        tag, d["ms_per_step"], d["value"], r.get("us_per_launch", 0), r.get("row_restricted_last_layer_us") or 0, r.get("tiles"),
        h.get("us_per_launch", 0)))
