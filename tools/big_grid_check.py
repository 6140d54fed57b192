#!/usr/bin/env python3
"""Grids beyond 2^32 cells: the interior-kernel families (blocked pairs, lean single steps, barrier-free, unfused) must
agree bit for bit on EVERY cell and on the receivers (the oracle is too slow there).  The fields start from seeded random
data, so every cell is live from step 0.   usage: tools/big_grid_check.py [--fcc] [--double] [Nx Ny Nz] [Nt]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pffdtd_amd import engine, sim_data, synth  # noqa: E402

fcc = "--fcc" in sys.argv  # 13-point, folded: the sizes are the STORED grid
prec = "double" if "--double" in sys.argv else "single"
argv = [a for a in sys.argv if a not in ("--fcc", "--double")]
n = [int(v) for v in argv[1:4]] if len(argv) >= 4 else [2112, 1024, 1024]
Nt = int(argv[4]) if len(argv) > 4 else 12
t0 = time.time()
c = [v // 2 for v in n]
src = [n[0] - 40, c[1], c[2]]  # linear indices of the source / these receivers lie beyond 2^32
rcv = [[n[0] - 37, c[1], c[2] - 2], [n[0] - 9, n[1] - 10, n[2] - 11], [c[0] + 3, c[1], c[2] - 2], [6, 7, 8]]
if fcc:
    sim = synth.shoebox(n[0], 2 * (n[1] - 1), n[2], Nt=Nt, fcc=True, Nm=2, Mb=[11, 3], src=[src[0], c[1], c[2] + (src[0] + c[1] + c[2]) % 2],
                        rcv=[[r[0], r[1] // 2 + 8, r[2] + (r[0] + r[1] // 2 + 8 + r[2]) % 2] for r in rcv])
    synth.fold_fcc(sim)
    synth.sort_sim(sim)
else:
    sim = synth.shoebox(*n, Nt=Nt, Nm=2, Mb=[11, 3], src=src, rcv=rcv)
sd = sim_data.SimData.from_sim(sim, prec, build_mask=False)
assert [sd.Nx, sd.Ny, sd.Nz] == n, (sd.Nx, sd.Ny, sd.Nz)
sd.scale_input()
print(f"scene {n} {prec} = {np.prod(n)/2**32:.2f} x 2^32 cells, Nb={sd.Nb}, built in {time.time()-t0:.1f}s", flush=True)
rb = 4 if prec == "single" else 8
tdt = torch.float32 if rb == 4 else torch.float64
P = engine.grid_pitch(n[2], rb)
shape = (n[0], n[1] * P)
gen = torch.Generator(device="cuda")
gen.manual_seed(11)
init = [((torch.rand(shape, generator=gen, device="cuda", dtype=tdt) * 2 - 1) * 1e-3) for _ in range(2)]
g = [torch.empty(shape, dtype=tdt, device="cuda") for _ in range(2)]
ref_out, ref_g = None, None
# 40 = blocked pairs forced, 0 = what the engine picks, 4 = barrier-free kernel with virtual ghosts, 25 = lean fused kernel
# (7-point), 3 = the reference's kernel sequence (memory flips, marching kernel, ABC list kernels)
for v in ((40, 0, 4, 3) if fcc else (40, 0, 25, 4, 3)):
    for a, b in zip(g, init):
        a.copy_(b)
    sd.u_out[:] = 0
    t0 = time.time()
    eng = engine.HipEngine(sd, air_variant=v, timing=True, ext_u0=g[0].data_ptr(), ext_u1=g[1].data_ptr())
    eng.run(0, Nt)
    eng.sync()
    tm = eng.timing()
    eng.close()
    out = sd.u_out.copy()
    print(f"variant {v}: {time.time()-t0:.1f}s, blocked launches {tm['tb2_launches']}, peak |out| {np.abs(out).max():.3e}", flush=True)
    if v == 40:
        assert tm["tb2_launches"] > 0, "pairs were forced but the two-steps-per-pass kernel never ran"
    view = [t.view(n[0], n[1], P)[1:-1, 1:-1, 1:n[2] - 1] for t in g]
    if ref_out is None:
        ref_out, ref_g = out, [t.clone() for t in g]
        assert np.abs(out).max() > 0 and all(bool(torch.isfinite(t).all()) for t in view)
    else:
        assert np.array_equal(out, ref_out), f"variant {v}: receivers differ, max|d|={np.abs(out-ref_out).max()}"
        for t, r in zip(view, ref_g):
            assert bool(torch.equal(t, r.view(n[0], n[1], P)[1:-1, 1:-1, 1:n[2] - 1])), f"variant {v}: fields differ"
print("big-grid check OK: all kernel families agree bit for bit on every cell")
