#!/usr/bin/env python3
"""Experiment: what do the floor / ceiling (z-normal) wall nodes cost in k_boundary at 1024^3?  Times the step with
and without them (physics of the second scene is meaningless)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from pffdtd_amd import engine, sim_data, synth

n = 1024
sim = synth.shoebox(n, n, n, Nt=40, Nm=1, Mb=11, lossy=True)
for drop in (False, True):
    s2 = {k: dict(v) for k, v in sim.items()}
    if drop:
        v = s2["vox_out"]
        ii = v["bn_ixyz"]
        iz, iy, ix = ii % n, (ii // n) % n, ii // (n * n)
        inner = (ix > 4) & (ix < n - 5) & (iy > 4) & (iy < n - 5)
        keep = ~(inner & ((iz <= 4) | (iz >= n - 5)))
        for k in ("bn_ixyz", "adj_bn", "mat_bn", "saf_bn"):
            v[k] = v[k][keep]
        v["Nb"] = np.int64(keep.sum())
    sd = sim_data.SimData.from_sim(s2, "single", build_mask=False)
    sd.scale_input()
    eng = engine.HipEngine(sd, air_variant=25, timing=True)
    eng.run(0, 5); eng.sync(); eng.timing(reset=True)
    t0 = time.perf_counter(); eng.run(5, 30); eng.sync(); el = (time.perf_counter() - t0) / 30
    tm = eng.timing()
    print(f"drop_zwalls={drop}: Nb={sd.Nb} Nbl={sd.Nbl} step {el*1e3:.3f} ms, air {tm['air_ms_total']/tm['steps']:.3f} ms, rest {el*1e3 - tm['air_ms_total']/tm['steps']:.3f} ms", flush=True)
    eng.close()
