#!/usr/bin/env python3
"""Summarise a tools/prof_air.sh output directory: per kernel average duration and PMC counters."""
import collections
import csv
import glob
import sys

out = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "pf::"
dur = {}
for f in glob.glob(f"{out}/stats/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            dur[r["Name"][:48]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{out}/pmc*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            ctr[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            ctr[r["Kernel_Name"][:48]]["_VGPR"] = [float(r["VGPR_Count"])]
            ctr[r["Kernel_Name"][:48]]["_LDS"] = [float(r["LDS_Block_Size"])]
for k in sorted(set(dur) | set(ctr)):
    c, us = dur.get(k, (0, float("nan")))
    print(f"== {k}  calls={c} avg={us:.1f} us")
    for name, vals in sorted(ctr.get(k, {}).items()):
        v = sorted(vals)
        med = v[len(v) // 2]
        extra = ""
        if name == "FETCH_SIZE":
            extra = f"  (x2 gfx950 correction: {med*2*1024/1e9:.3f} GB)"
        if name == "WRITE_SIZE":
            extra = f"  ({med*1024/1e9:.3f} GB)"
        print(f"     {name:28s} {med:16.1f}{extra}")
