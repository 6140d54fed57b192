#!/usr/bin/env python3
"""Kernel timeline of one steady-state pass from a rocprofv3 --kernel-trace directory: every kernel between two consecutive
launches of the dominant kernel (default k_tb3<..., false, 3>), with start offset, duration and stream/queue.
usage: timeline.py <trace dir> [anchor substring] [which occurrence from the end]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "false, 3>"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = []
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_tb3<" in r["Kernel_Name"] and anchor in r["Kernel_Name"]]
if len(idx) < back + 1:
    sys.exit(f"only {len(idx)} anchor launches")
i0, i1 = idx[-back - 1], idx[-back]
t0 = int(rows[i0]["Start_Timestamp"])
print(f"pass = {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us between two launches of the anchor")
for r in rows[i0:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void pf::", "")
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3} grid {r.get('Grid_Size', '?'):>8}  {name[:90]}")
