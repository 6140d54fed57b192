#!/usr/bin/env python3
"""Access-pattern calibration: linear 2R+1W stream vs the marching tile pattern of the air kernels (no stencil)."""
import functools
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pffdtd_amd import build  # noqa: E402

print = functools.partial(print, flush=True)
L = build.load_probe()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
a = torch.rand((n, n * n), device="cuda") * 1e-3
b = torch.rand((n, n * n), device="cuda") * 1e-3
byt = 3 * 4 * (n - 2) * (n - 2) * n


def run(kind, R=4, WY=4, PF=1, chunk=128, sw=1, reps=10):
    ms = L.pf_membench(a.data_ptr(), b.data_ptr(), n, n, n, kind, R, WY, PF, chunk, sw, reps)
    return ms


for mode, unr in [(0, 1), (1, 1), (2, 1), (3, 1), (4, 1), (7, 1), (0, 4), (3, 4), (4, 4), (7, 4), (6, 4), (5, 4), (4, 2), (7 + 8 * 3, 4), (7 + 8 * 6, 4), (7 + 8 * 8, 4), (7 + 8 * 10, 4), (4 + 8 * 3, 4), (4 + 8 * 8, 4)]:
    ms = run(0, mode, unr)
    print(f"linear mode={mode} (nt-ld {mode&1}, nt-st {(mode>>1)&1}, one-shot {(mode>>2)&1}, streams {1<<(mode>>3)}) unroll={unr}  {ms:7.3f} ms  {3*4*n**3/ms/1e6:7.1f} GB/s")
for m in range(1, 8):
    ms = run(1 + m, 4, 4, 1, 32)
    print(f"march R=4 WY=4 PF=1 chunk=32 nt(u1 ld {m&1}, u0 ld {(m>>1)&1}, st {(m>>2)&1})  {ms:7.3f} ms  {byt/ms/1e6:7.1f} GB/s")
for kind, R, WY in [(20, 4, 1), (27, 4, 1), (20, 4, 2), (27, 4, 2), (20, 2, 2), (27, 2, 2), (27, 1, 4), (27, 8, 1)]:
    for chunk in (4, 16, 64, 256):
        ms = run(kind, R, WY, 1, chunk)
        print(f"march full-row tiles (WZ=4) R={R} WY={WY} nt={kind-20} chunk={chunk:4d}  {ms:7.3f} ms  {byt/ms/1e6:7.1f} GB/s")
if len(sys.argv) > 2:
    sys.exit(0)
for (R, WY, PF) in [(4, 4, 1), (4, 4, 2), (4, 4, 3), (2, 4, 1), (2, 4, 2), (2, 4, 4), (1, 4, 2), (1, 4, 4), (1, 4, 8), (2, 8, 2),
                    (4, 8, 2), (8, 4, 1), (8, 4, 2), (1, 8, 4)]:
    for chunk in (32, 128):
        ms = run(1, R, WY, PF, chunk)
        print(f"march R={R} WY={WY} PF={PF} chunk={chunk:4d}  {ms:7.3f} ms  {byt/ms/1e6:7.1f} GB/s" if ms > 0 else f"march R={R} WY={WY} PF={PF}: unsupported")
