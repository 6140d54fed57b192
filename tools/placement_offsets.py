#!/usr/bin/env python3
"""Round-5 experiment (verdict item: close the placement question): the four streams of k_tb3 carved from ONE allocation, with a
sweep of 2 MiB-multiple gaps between consecutive grids, against four separate allocations -- repeated over fresh allocations.
Prints ms per launch of k_tb3<float, 3, 8> (banded order, chunk 32) on a 1024^3 free-field grid for every layout.
usage: placement_offsets.py [n] [repeats]"""
import ctypes
import functools
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pffdtd_amd import build, engine  # noqa: E402

print = functools.partial(print, flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = build.load_probe()
vp, i32, i64, d = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
L.pf_tb3_probe.restype = d
L.pf_tb3_probe.argtypes = [vp, vp, vp, vp, i64, i64, i64, d, d, i32, i32, i32, i32]
P = engine.grid_pitch(n, 4)
gb = n * n * P * 4
MiB2 = 2 << 20


def time_layout(ptrs):
    return L.pf_tb3_probe(ptrs[0], ptrs[1], ptrs[2], ptrs[3], n, n, n, 1.0, 0.1, 8, 10308, 32, 4)


for rep in range(reps):
    # (a) four separate allocations, as hipMalloc hands them out
    sep = [torch.zeros(gb // 4, dtype=torch.float32, device="cuda") for _ in range(4)]
    t_sep = time_layout([g.data_ptr() for g in sep])
    del sep
    torch.cuda.empty_cache()
    # (b) one allocation, grids at multiples of (grid + gap), gap = k x 2 MiB
    maxgap = 16
    big = torch.zeros((4 * (gb + maxgap * MiB2) + MiB2) // 4, dtype=torch.float32, device="cuda")
    base = (big.data_ptr() + MiB2 - 1) // MiB2 * MiB2
    row = []
    for k in range(maxgap + 1):
        stride = gb + k * MiB2
        row.append(time_layout([base + i * stride for i in range(4)]))
    # the same with the roles permuted (read grids 0, 2; write 1, 3) at gap 0
    t_perm = time_layout([base, base + 2 * gb, base + gb, base + 3 * gb])
    del big
    torch.cuda.empty_cache()
    print(f"allocation {rep}: separate {t_sep:.3f} ms | one allocation, gap k x 2 MiB, k = 0..{maxgap}: " + " ".join(f"{t:.3f}" for t in row)
          + f" | roles interleaved at gap 0: {t_perm:.3f}")
