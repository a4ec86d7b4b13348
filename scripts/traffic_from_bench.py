"""profiles/<round>/traffic_<workload>_d<d>.json from a bench line whose traffic was MEASURED IN THE RUN (bench.py --pmc):
what bench.py falls back to (labelled) on a box without rocprofv3.  usage: python scripts/traffic_from_bench.py <bench.json> <round>"""
import json
import os
import sys

line, rnd = json.load(open(sys.argv[1])), sys.argv[2]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", rnd)
r = line["roofline"]
free = r.get("hbm_reuse_free")
if isinstance(free, dict) and free.get("access_pattern") == "random":  # (round 6: the permuted graph at the top level, the strided twin inside)
    legs = {"synth-10M": r.get("hbm_bound"), "regular-15M-perm": free, "regular-15M": free.get("strided"), "synth-1M": r.get("cache_boundary")}
else:
    legs = {"synth-10M": r.get("hbm_bound"), "regular-15M": free, "synth-1M": r.get("cache_boundary")}
legs = {k: v for k, v in legs.items() if isinstance(v, dict)}
legs[line["config"]["workload"].split("-shape")[0].split(":")[0].split(" ")[0]] = r  # the headline's own graph
for c in line.get("configs") or []:
    if isinstance(c.get("roofline"), dict) and "bytes_gather" in c["roofline"]:
        legs.setdefault(c["graph"], c["roofline"])
for name, leg in legs.items():
    if not isinstance(leg, dict) or not str(leg.get("traffic_source", "")).startswith("measured in this run"):
        continue
    d = int(line["config"]["dim"]) if leg is r else 64
    out = {"workload": name, "d": d, "hbm_bytes_per_launch": leg["traffic"], "l2_hit_rate": leg.get("traffic_l2_hit_rate"),
           "us_per_launch": leg["us_per_launch"], "hbm_side_gbs": leg["traffic_gbs"], "frac_of_8TBs_hbm_side": leg["frac_traffic"],
           "bytes_gather": leg.get("bytes_gather"), "method": leg["traffic_source"], "from": os.path.basename(sys.argv[1])}
    path = os.path.join(root, "traffic_%s_d%d.json" % (name, d))
    json.dump(out, open(path, "w"), indent=1)
    print(path, "%.3f of the HBM peak at the L2s' memory side" % out["frac_of_8TBs_hbm_side"])
