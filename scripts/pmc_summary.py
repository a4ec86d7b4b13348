"""Aggregate rocprofv3 --pmc CSVs (one directory per pass) into per-kernel averages per dispatch."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        out[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {}
for k, ctrs in out.items():
    summary[k] = {c: {"avg": sum(v) / len(v), "n": len(v)} for c, v in ctrs.items()}
keep = {k: v for k, v in summary.items() if any(s in k for s in ("spmm", "adam", "lincomb", "bpr", "score", "topk"))}
print(json.dumps(keep, indent=1, sort_keys=True))
