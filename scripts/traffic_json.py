"""Turn a scripts/pmc.sh summary into profiles/traffic_<workload>_d<d>.json (read by bench.py's
roofline.traffic).  usage: python scripts/traffic_json.py gpurun_out/pmc_<tag>/summary.json <workload> <d> [round dir] [kernel_stats.csv]

kernel_stats.csv: the `rocprofv3 --kernel-trace --stats` CSV of the SAME bench command WITHOUT counters (scripts/prof.sh;
committed beside this file): the dominant kernel's average launch duration goes into `us_per_launch`, so that
bytes / time = the fraction of the HBM peak follows from tracked files alone.

HBM-side bytes per launch of the dominant kernel = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md,
HBM/rocprofv3 section: gfx950 tallies its 128-byte read requests as 64 B; WRITE_SIZE is exact).  The
factor is re-checked in the same run on adam_kernel, whose traffic is known: 16 B read and 12 B
written per element (run the PMC passes with bench.py --separate-adam so that kernel exists)."""
import json
import os
import sys

src, workload, d = sys.argv[1], sys.argv[2], int(sys.argv[3])
s = json.load(open(src))
dense = [k for k in s if k.startswith("spmm_tile_kernel<%d, 1, 8, true" % (d // 4)) and "FETCH_SIZE" in s[k]]
# the plain (EPI = 0) instantiation is the dominant kernel; with --separate-adam it is the only dense one
dense.sort(key=lambda k: -s[k]["FETCH_SIZE"]["n"])
k = dense[0]
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KILOBYTES (round 3: the earlier "x 1024 below 1e7" guess broke on launches
# that move more than 10 GB)
fetch, write = s[k]["FETCH_SIZE"]["avg"] * 1024, s[k]["WRITE_SIZE"]["avg"] * 1024
hit, miss = s[k].get("TCC_HIT_sum", {}).get("avg"), s[k].get("TCC_MISS_sum", {}).get("avg")
out = {
    "kernel": k, "workload": workload, "d": d,
    "fetch_size_bytes_raw": fetch, "write_size_bytes": write, "hbm_bytes_per_launch": 2 * fetch + write,
    "l2_hit_rate": (hit / (hit + miss)) if hit is not None and miss is not None else None,
    "dispatches_averaged": s[k]["FETCH_SIZE"]["n"],
}
adam = s.get("adam_kernel")
if adam and "FETCH_SIZE" in adam:
    af = adam["FETCH_SIZE"]["avg"] * 1024
    aw = adam["WRITE_SIZE"]["avg"] * 1024
    out["calibration_adam_kernel"] = {"fetch_raw": af, "write": aw,
                                      "note": "known traffic: reads = 4/3 x writes (p,g,m,v in; p,m,v out)",
                                      "reads_over_writes_with_x2": 2 * af / aw}
if len(sys.argv) > 5:
    import csv

    rows = [r for r in csv.DictReader(open(sys.argv[5]))
            if k.replace(" ", "") in r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace(" ", "")]
    if rows:
        r = max(rows, key=lambda r: int(r["Calls"]))
        us = float(r["AverageNs"]) / 1e3
        out["us_per_launch"] = us
        out["us_per_launch_source"] = "%s: %s calls, rocprofv3 --kernel-trace --stats of the same command without counters" \
                                      % (os.path.basename(sys.argv[5]), r["Calls"])
        out["hbm_side_gbs"] = out["hbm_bytes_per_launch"] / us / 1e3
        out["frac_of_8TBs_hbm_side"] = out["hbm_side_gbs"] / 8000.0
out["method"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum, one counter set per pass "
                 "(scripts/pmc.sh, bench.py --separate-adam), per-dispatch average; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                 "(gfx950 tallies 128-B read requests at 64 B), checked on adam_kernel in the same run. Fabric-side bytes: "
                 "Infinity Cache hits are included, so for a cache-resident panel this is not DRAM traffic.")
sub = sys.argv[4] if len(sys.argv) > 4 else "r04"  # profiles/<round>/: bench.py reads the newest round's file
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", sub, "traffic_%s_d%d.json" % (workload, d))
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
