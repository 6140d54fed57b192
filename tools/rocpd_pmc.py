#!/usr/bin/env python3
"""Per-kernel medians of the PMC counters in rocprofv3 rocpd databases.
usage: tools/rocpd_pmc.py <kernel-name-substring> <run_results.db> [more.db ...]"""
import collections
import re
import sqlite3
import statistics
import sys

pat = sys.argv[1]
for f in sys.argv[2:]:
    cur = sqlite3.connect(f).cursor()
    rows = cur.execute("select kernel_name, counter_name, value, duration, vgpr_count, accum_vgpr_count, scratch_size, dispatch_id from counters_collection").fetchall()
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for nm, cn, val, dur, vg, ag, sc, did in rows:
        if pat not in nm:
            continue
        short = re.sub(r"\(.*", "", nm).replace("void pf::", "")
        per[short][cn].append(val)
        per[short]["_duration_us"].append(dur / 1e3)
        meta[short] = (vg, ag, sc)
    for k, d in per.items():
        print(f"== {k}  vgpr={meta[k][0]} agpr={meta[k][1]} scratch={meta[k][2]}  ({f})")
        for cn, vals in sorted(d.items()):
            print(f"     {cn:24s} median {statistics.median(vals):16.1f}   n={len(vals)}")
