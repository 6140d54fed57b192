#!/usr/bin/env python3
"""Print the per-stream kernel timeline of a few consecutive steps from a rocprofv3 --kernel-trace csv
(usage: tools/timeline.py <dir> [first_air_index] [n_kernels])."""
import csv
import glob
import sys

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 600
count = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows = []
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:skip + count]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f} us  q{r.get('Queue_Id','?'):>3} grid {r.get('Grid_Size','?'):>9}  {r['Kernel_Name'][:70]}")
