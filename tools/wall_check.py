#!/usr/bin/env python3
"""Where do the wall-region pairs (pf_wall.h) and the single-step engine differ?  Runs a forced-pair engine and the lean
single-step engine on the same scene and prints, per grid and per region of the shell, how many cells differ and the
first few coordinates -- the debugging aid behind tests/test_hip_tb2.py."""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
from pffdtd_amd import engine, sim_data, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs=3, default=[36, 64, 280])
ap.add_argument("--nt", type=int, default=8)
ap.add_argument("--prec", default="single")
ap.add_argument("--wall", type=int, default=3)
ap.add_argument("--mb", type=int, nargs="+", default=[11, 3])
ap.add_argument("--rigid", action="store_true")
ap.add_argument("--src", type=int, nargs=3, default=None)
ap.add_argument("--random", action="store_true", help="seeded random initial fields (every cell live from step 0)")
ap.add_argument("--debug", type=lambda s: int(s, 0), default=0)
a = ap.parse_args()

n = a.n
sim = synth.shoebox(*n, Nt=a.nt, Nm=len(a.mb), Mb=a.mb, src=a.src, wall=a.wall, lossy=not a.rigid)


def run(variant, debug=0):
    sd = sim_data.SimData.from_sim(sim, a.prec, build_mask=False)
    sd.scale_input()
    eng = engine.HipEngine(sd, air_variant=variant, timing=True, debug=debug)
    if a.random:
        rng = np.random.default_rng(5)
        dt = np.float32 if a.prec == "single" else np.float64
        for k in (0, 1):
            eng.set_grid(k, (rng.standard_normal(n) * 1e-2).astype(dt))
    eng.run(0, sd.Nt)
    tm = eng.timing()
    g = [eng.get_grid(0).copy(), eng.get_grid(1).copy()]
    eng.close()
    return sd.u_out.copy(), g, tm


os.environ.setdefault("PFFDTD_VERBOSE", "1")
out1, g1, tm1 = run(40, a.debug)
out0, g0, tm0 = run(25)
print("pairs:", tm1["tb2_launches"], "launches; receivers equal:", np.array_equal(out0, out1))
ok = True
for which in (0, 1):
    d = g0[which][1:-1, 1:-1, 1:-1] != g1[which][1:-1, 1:-1, 1:-1]
    print(f"grid {which}: {int(d.sum())} of {d.size} interior cells differ; max |u| = {np.abs(g0[which]).max():.3e}")
    if d.any():
        ok = False
        idx = np.argwhere(d) + 1
        for ax, nm in enumerate("xyz"):
            vals, cnt = np.unique(idx[:, ax], return_counts=True)
            order = np.argsort(-cnt)[:12]
            print(f"   {nm}: " + ", ".join(f"{vals[i]}:{cnt[i]}" for i in sorted(order)))
        for p in idx[:10]:
            print("   ", tuple(p), g0[which][tuple(p)], g1[which][tuple(p)])
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
