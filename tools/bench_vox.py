#!/usr/bin/env python3
"""Voxelizer measurement: device time of the binning and ray-triangle kernels on the reference's CTK / Musikverein
exports at their test-script resolutions, next to the numpy oracle (CPU, one core) on a bounded sample.

    python tools/bench_vox.py [--reps 5] [--no-cpu]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
from pffdtd_amd import scenes, setup_io  # noqa: E402
from pffdtd_amd.room_geo import RoomGeo  # noqa: E402
from pffdtd_amd.voxelizer import cut_legs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--no-cpu", action="store_true")
a = ap.parse_args()

out = []
for name in ("ctk_cart_viz", "ctk_cart_gpu", "mv_fcc_viz", "mv_fcc_gpu"):
    cfg = scenes.CONFIGS[name]
    rg = RoomGeo(str(scenes.model_path(cfg["model"])))
    sc = setup_io.SimConsts(Tc=cfg["Tc"], rh=cfg["rh"], fmax=cfg["fmax"], PPW=cfg["PPW"], fcc=cfg["fcc_flag"])
    cg = setup_io.CartGrid(h=sc.h, offset=3.5, bmin=rg.bmin, bmax=rg.bmax, fcc=cfg["fcc_flag"])
    best = None
    for _ in range(a.reps):
        t0 = time.perf_counter()
        bn, adj, tidx, nd, st = cut_legs(cg.xv, cg.yv, cg.zv, sc.h, cfg["fcc_flag"], rg.tris_pre)
        st["wall_s"] = time.perf_counter() - t0
        if best is None or st["ms_vox"] + st["ms_bin"] < best["ms_vox"] + best["ms_bin"]:
            best = st
    NN = 12 if cfg["fcc_flag"] else 6
    live = 0.5 if cfg["fcc_flag"] else 1.0
    r = {"config": name, "grid": [cg.Nx, cg.Ny, cg.Nz], "ntris": int(rg.tris.shape[0]), "Nb": int(bn.size),
         "ms_bin": round(best["ms_bin"], 3), "ms_vox": round(best["ms_vox"], 3), "wall_s": round(best["wall_s"], 3),
         "cells_nonempty": best["ncells_nonempty"], "pairs": best["npairs"],
         # upper bound on ray tests: every live point of a cell against every candidate triangle of that cell, NN legs
         "ray_slots_per_s": round(best["npairs"] * 256 * live * NN / (best["ms_vox"] * 1e-3) / 1e9, 2),
         "grid_points_per_s_G": round(cg.Npts / ((best["ms_vox"] + best["ms_bin"]) * 1e-3) / 1e9, 1)}
    out.append(r)
    print(json.dumps(r), flush=True)

if not a.no_cpu:
    import vox_oracle as vo
    cfg = scenes.CONFIGS["ctk_cart_viz"]
    rg = RoomGeo(str(scenes.model_path("CTK")))
    sc = setup_io.SimConsts(Tc=20, rh=50, fmax=cfg["fmax"], PPW=cfg["PPW"], fcc=False)
    cg = setup_io.CartGrid(h=sc.h, offset=3.5, bmin=rg.bmin, bmax=rg.bmax, fcc=False)
    t0 = time.perf_counter()
    bn, *_ = vo.calc_adj(cg.xv, cg.yv, cg.zv, sc.h, False, rg.tris_pre)
    el = time.perf_counter() - t0
    print(json.dumps({"cpu_baseline": {"kind": "port", "what": "oracle/vox_oracle.py (numpy, 1 core) on ctk_cart_viz 234x154x85",
                                       "seconds": round(el, 2), "Nb": int(bn.size),
                                       "note": "the reference voxelizer itself: 12.5 s (this grid), 101.5 s (ctk_cart_gpu grid), 436 s "
                                               "(mv_fcc_viz grid) on one core of the build container (tests/golden/make_golden_vox.py)"}}), flush=True)
