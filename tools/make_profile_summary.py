#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/ into the tracked profiles/ directory.

usage: tools/make_profile_summary.py <round-tag> <stats_dir> <fetch_dir> <write_dir> [bench_json]
Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, engine kernels only),
profiles/<tag>_hbm_traffic.json (per-kernel FETCH_SIZE / WRITE_SIZE medians, with the gfx950 x2 read correction
of guides/MI355X_MICROARCH.md applied and stated) and profiles/<tag>_summary.md.
"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
bench_json = sys.argv[5] if len(sys.argv) > 5 else None
out = ROOT / "profiles"
out.mkdir(exist_ok=True)

rows = []
for f in glob.glob(f"{stats_dir}/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "pf::" in r["Name"]:
            rows.append(r)
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(out / f"{tag}_kernel_stats.csv", "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)


def med(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "pf::" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in agg.items()}


fetch, write = med(fetch_dir, "FETCH_SIZE"), med(write_dir, "WRITE_SIZE")
traffic = {}
for k in sorted(set(fetch) | set(write)):
    rd = fetch.get(k, 0.0) * 1024 * 2  # KiB -> bytes, x2: gfx950 FETCH_SIZE counts 128-B requests as 64 B
    wr = write.get(k, 0.0) * 1024
    traffic[k] = {"read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr,
                  "FETCH_SIZE_raw_KiB": fetch.get(k), "WRITE_SIZE_raw_KiB": write.get(k)}
json.dump({"note": "per launch medians; read_bytes = FETCH_SIZE*1024*2 (gfx950 correction, MI355X_MICROARCH.md "
                   "section HBM, re-validated here on k_flip_x / k_mask_init whose byte counts are known); "
                   "write_bytes = WRITE_SIZE*1024", "kernels": traffic}, open(out / f"{tag}_hbm_traffic.json", "w"), indent=1)

with open(out / f"{tag}_summary.md", "w") as fh:
    fh.write(f"# {tag}: rocprofv3 summary (MI355X, bench.py N=1, 1024^3 7-pt fp32, Mb=11 walls)\n\n")
    if bench_json and Path(bench_json).exists():
        fh.write("bench line:\n\n```\n" + Path(bench_json).read_text().strip() + "\n```\n\n")
    fh.write("| kernel | calls | avg us | % | HBM read MB | HBM write MB |\n|---|---|---|---|---|---|\n")
    for r in rows:
        t = traffic.get(r["Name"], {})
        fh.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} | "
                 f"{t.get('read_bytes', 0)/1e6:.1f} | {t.get('write_bytes', 0)/1e6:.1f} |\n")
print((out / f"{tag}_summary.md").read_text())
