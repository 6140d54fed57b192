#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (run_results.db): calls, average / min / max duration, share.
usage: tools/rocpd_stats.py <run_results.db> [csv-out]"""
import csv
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), avg(end - start), min(end - start), max(end - start), sum(end - start) from kernels group by {name_col}").fetchall()
tot = sum(r[5] for r in rows) or 1
rows.sort(key=lambda r: -r[5])
out = []
for nm, n, avg, mn, mx, sm in rows:
    short = re.sub(r"\(.*", "", nm).replace("void pf::", "")
    out.append({"Name": short, "Calls": n, "AverageNs": round(avg, 1), "MinNs": mn, "MaxNs": mx, "TotalDurationNs": sm, "Percentage": round(100 * sm / tot, 2)})
for r in out[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f'{r["Name"][:84]:84s} calls={r["Calls"]:6d} avg={r["AverageNs"]/1e3:9.1f} us  min={r["MinNs"]/1e3:9.1f} max={r["MaxNs"]/1e3:9.1f}  {r["Percentage"]:5.1f} %')
if len(sys.argv) > 2 and sys.argv[2] != "-":
    with open(sys.argv[2], "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(out[0].keys()))
        w.writeheader()
        w.writerows(out)
