#!/usr/bin/env python3
"""Sweep air-kernel tile variants / chunk lengths on one GPU and print a table (used for tuning, not a test)."""
import argparse
import functools
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
print = functools.partial(print, flush=True)

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--nx", type=int, default=0, help="planes along x (default: size)")
ap.add_argument("--ny", type=int, default=0)
ap.add_argument("--nz", type=int, default=0)
ap.add_argument("--precision", default="single")
ap.add_argument("--fcc", action="store_true")
ap.add_argument("--rigid", action="store_true")
ap.add_argument("--variants", default="0,3,4,7,25,40")
ap.add_argument("--chunks", default="0")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--numerics", type=int, default=0)
ap.add_argument("--debug", default="0")
args = ap.parse_args()

import torch  # noqa: E402
from pffdtd_amd import engine, sim_data, synth  # noqa: E402

n = args.size
nx = args.nx or n
t0 = time.time()
if args.fcc:
    sim = synth.shoebox(nx, 2 * (n - 1), n, Nt=args.steps + 3, fcc=True, Nm=1, Mb=11, lossy=not args.rigid)
    synth.fold_fcc(sim)
    synth.sort_sim(sim)
else:
    sim = synth.shoebox(nx, args.ny or n, args.nz or n, Nt=args.steps + 3, Nm=1, Mb=11, lossy=not args.rigid)
sd = sim_data.SimData.from_sim(sim, args.precision, build_mask=False)
sd.scale_input()
print(f"scene {sd.Nx}x{sd.Ny}x{sd.Nz} built in {time.time()-t0:.1f}s Nb={sd.Nb} Nbl={sd.Nbl} Nba={sd.Nba}")
rb = sd.real_bytes
P = engine.grid_pitch(sd.Nz, rb)
tdt = torch.float32 if rb == 4 else torch.float64
grids = [torch.zeros((sd.Nx, sd.Ny * P), dtype=tdt, device="cuda") for _ in range(2)]
bpv = 3 * rb + 0.125
upd = (sd.Nx - 2) * (sd.Ny - 2) * (sd.Nz - 2)
print(f"{'var':>4} {'chunk':>5} {'air ms':>8} {'air GB/s':>9} {'frac8T':>7} {'step ms':>8} {'Gvox/s':>8}")
for v in [int(x) for x in args.variants.split(",")]:
  for dbg in [int(x) for x in args.debug.split(",")]:
    for c in [int(x) for x in args.chunks.split(",")]:
        for g in grids:
            g.copy_((torch.rand(g.shape, device="cuda", dtype=torch.float32) * 2 - 1) * 1e-3)
        torch.cuda.synchronize()
        try:
            eng = engine.HipEngine(sd, air_variant=v, air_chunk=c, timing=True, numerics=args.numerics, debug=dbg,
                                   ext_u0=grids[0].data_ptr(), ext_u1=grids[1].data_ptr())
            eng.run(0, 3)
            eng.timing(reset=True)
            t0 = time.perf_counter()
            eng.run(3, args.steps)
            eng.sync()
            el = time.perf_counter() - t0
            tm = eng.timing()
            eng.close()
        except Exception as ex:  # noqa: BLE001
            print(f"{v:4d} {c:5d} EXC {ex}")
            continue
        air = tm["air_ms_total"] / tm["air_launches"]
        print(f"{v:4d}/{dbg:<2d} {c:5d} {air:8.3f} {upd*bpv/air/1e6:9.1f} {upd*bpv/air/1e6/8000:7.3f} {el/args.steps*1e3:8.3f} "
              f"{sd.Npts*args.steps/el/1e9:8.2f}")
