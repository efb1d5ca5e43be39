#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs of tools/prof_passes.sh into one JSON: per kernel, the mean
per-launch value of every counter, plus kernel-trace statistics."""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
out = {"counters_per_launch_mean": {}, "kernel_stats": []}
for f in sorted(glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "trace_kernel" not in k:
            continue
        k = k.replace("void nnbvh::", "").split("(")[0]
        agg[(k, r["Grid_Size"], r["VGPR_Count"], r["LDS_Block_Size"])][r["Counter_Name"]].append(
            float(r["Counter_Value"]))
    for (k, grid, vgpr, lds), cs in agg.items():
        d = out["counters_per_launch_mean"].setdefault(k, {"grid": grid, "vgpr": vgpr, "lds": lds})
        d.update({c: sum(v) / len(v) for c, v in cs.items()})
        d["launches_profiled"] = len(next(iter(cs.values())))
for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")):
    out["kernel_stats"] = list(csv.DictReader(open(f)))
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(out, open(dst, "w"), indent=1)
for k, d in out["counters_per_launch_mean"].items():
    print(k, json.dumps(d))
