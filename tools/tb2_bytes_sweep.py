#!/usr/bin/env python3
"""Pair kernel k_tb2_reg over tile heights and x-chunk lengths on a free-field 1024^3 grid (tools/libpf_probe.so): time per
launch, and -- run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv` -- the bytes each variant fetches
(tools/tb2_bytes_table.py joins the two).  usage: tb2_bytes_sweep.py [n] [tye,tye,...] [chunk,chunk,...] [reps]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pffdtd_amd import build, engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tyes = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "11304,11308,11316,11404,11208").split(",")]
chunks = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "8,16,32,64").split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
L = build.load_probe()
P = engine.grid_pitch(n, 4)
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
g = [(torch.rand((n, n * P), generator=gen, device="cuda") * 2 - 1) * 1e-3 for _ in range(4)]
S = [torch.cat([g[0], g[1]]), torch.cat([g[2], g[3]])]  # interleaved layouts (tye + 100000 rows, + 200000 planes): two storages
m = 8
cells = (n - 2 * m) ** 3
for tye in tyes:
    for chunk in chunks:
        lay = tye // 100000
        off = 4 * (P if lay == 1 else n * P)
        ptrs = [g[i].data_ptr() for i in range(4)] if lay == 0 else [S[0].data_ptr(), S[0].data_ptr() + off, S[1].data_ptr(), S[1].data_ptr() + off]
        ms = L.pf_tb2_probe(*ptrs, n, n, n, 0.5, 0.25, m, tye, chunk, reps)
        if ms < 0:
            print("probe failed:", L.pf_probe_last_error().decode(), flush=True)
            continue
        print(f"tye={tye} chunk={chunk} ms={ms:.4f} compulsory_GB={cells*16/1e9:.3f} algorithmic_TBps={cells*16/ms/1e9:.3f}", flush=True)
