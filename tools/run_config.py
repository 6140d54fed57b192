#!/usr/bin/env python3
"""Build one of the reference's test-script configurations from its scene export at full resolution on this GPU box
(sim_setup: voxelizer on the device, GPU prep) and time the HIP engine on it.

    python tools/run_config.py ctk_cart_gpu --steps 400            # BASELINE configs[1]
    python tools/run_config.py mv_fcc_gpu --steps 200              # BASELINE configs[2]
    python tools/run_config.py ctk_cart_viz --precision double --energy   # BASELINE configs[0] on the GPU

Prints one JSON line (Gvoxel-updates/s by the reference's formula Npts*steps/time, cpu_engine.h:357).
"""
import argparse
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402,F401  (before the engine library: see INTEGRATION.md 3)
from pffdtd_amd import engine, scenes, sim_data  # noqa: E402
from pffdtd_amd.sim_setup import sim_setup  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("name", choices=sorted(scenes.CONFIGS))
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--precision", default="single", choices=["single", "double"])
ap.add_argument("--fmax-scale", type=float, default=1.0, help="scale fmax (grid spacing) for smaller trial runs")
ap.add_argument("--fmax", type=float, default=None, help="override fmax (Hz)")
ap.add_argument("--ppw", type=float, default=None, help="override points per wavelength")
ap.add_argument("--duration", type=float, default=None, help="override the simulated duration (s)")
ap.add_argument("--debug", type=lambda v: int(v, 0), default=0, help="internal PF_DBG_* switches (csrc/pf_debug.h), through the library's internal hook")
ap.add_argument("--variant", type=int, default=0, help="pf_opts.air_variant (0 = automatic, 40 = blocked pairs forced)")
ap.add_argument("--chunk", type=int, default=0, help="pf_opts.air_chunk (planes marched per workgroup; 0 = automatic)")
ap.add_argument("--energy", action="store_true", help="run all Nt steps with the energy diagnostic (double only)")
ap.add_argument("--keep", default=None, help="keep the sim folder here instead of a temp dir")
a = ap.parse_args()

tmp = tempfile.TemporaryDirectory()
base = Path(a.keep) if a.keep else Path(tmp.name)
mats = scenes.write_materials(base / "materials")
folder = base / a.name
kw = scenes.setup_kwargs(a.name, folder, mats, save_folder_gpu=folder, compress=0)
if a.fmax:
    kw["fmax"] = a.fmax
if a.ppw:
    kw["PPW"] = a.ppw
if a.duration:
    kw["duration"] = a.duration
kw["fmax"] *= a.fmax_scale
if a.precision == "single" and not kw["diff_source"]:
    kw["diff_source"] = True  # fp32 needs a differentiated input (fdtd_data.h:392)
t0 = time.perf_counter()
vs = sim_setup(**kw)
t_setup = time.perf_counter() - t0
res = {"config": a.name, "precision": a.precision, "setup_s": round(t_setup, 2), "voxelizer": vs.stats,
       "Nb": int(vs.bn_ixyz.size)}

sd = sim_data.SimData.from_folder(folder, a.precision, build_mask=False)
if not a.energy:
    sd.scale_input()
res.update(grid=[sd.Nx, sd.Ny, sd.Nz], Npts=int(sd.Npts), Nt=int(sd.Nt), fcc_flag=int(sd.fcc_flag), Nbl=int(sd.Nbl), Nba=int(sd.Nba))
if a.energy:
    eng = engine.HipEngine(sd, energy=True)
    eng.energy_cfg(sd.h, sd.c, sd.Ts, sd.DEF)
    H, El, Ei = np.zeros(sd.Nt), np.zeros(sd.Nt + 1), np.zeros(sd.Nt + 1)
    t0 = time.perf_counter()
    eng.run_energy(0, sd.Nt, H, El, Ei)
    eng.sync()
    el = time.perf_counter() - t0
    tot = H + El[:-1]
    bal = (tot - Ei[:-1]) / (2.0 ** np.floor(np.log2(np.maximum(tot, 1e-300))))
    res.update(steps=int(sd.Nt), seconds=round(el, 3), energy_balance_max=float(np.abs(bal[sd.Nt // 4:]).max()),
               gvox_per_s=round(sd.Npts * sd.Nt / el / 1e9, 3))
else:
    K, W = min(a.steps, sd.Nt - a.warmup), a.warmup
    # state grids (the engine's own: it also places them) pre-filled with seeded noise of the magnitude of a running simulation: every cell is live
    # from step 0 (receivers included -- the source's own wave needs thousands of steps to reach them at this resolution),
    # and the clocks see the data activity of a real run (all-zero fields clock higher)
    import torch
    rb = 4 if a.precision == "single" else 8
    P = engine.grid_pitch(sd.Nz, rb)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    from pffdtd_amd.dist import _DevMem
    eng = engine.HipEngine(sd, timing=True, debug=a.debug, air_variant=a.variant, air_chunk=a.chunk)
    (snx, sny, _), spitch, exchanged = eng.layout()  # as stored (rooms: the engine may keep the file's x and z axes exchanged)
    res["axes_exchanged"] = exchanged
    for p in eng.state_grids():
        g = torch.as_tensor(_DevMem(p, (snx, sny * spitch), "<f4" if rb == 4 else "<f8"), device="cuda")
        assert g.data_ptr() == p
        g.copy_((torch.rand(g.shape, generator=gen, device="cuda", dtype=g.dtype) * 2 - 1) * 1e-3)
    torch.cuda.synchronize()
    eng.run(0, W)
    eng.sync()
    eng.timing(reset=True)
    t0 = time.perf_counter()
    eng.run(W, K)
    eng.sync()
    el = time.perf_counter() - t0
    tm = eng.timing()
    res.update(steps=K, seconds=round(el, 4), ms_per_step=round(el / K * 1e3, 4), gvox_per_s=round(sd.Npts * K / el / 1e9, 2),
               air_ms_per_step=round(tm["air_ms_total"] / max(tm["steps"], 1), 4),
               finite=bool(np.isfinite(sd.u_out).all()), out_peak=float(np.abs(sd.u_out).max()),
               air_path=int(tm["air_path"]), tune_ms=[round(v, 4) for v in tm["tune_ms"]],
               blocked_cell_fraction=round(tm["tb2_cells"] / sd.Npts, 4) if tm["tb2_launches"] else 0.0)
eng.close()
print(json.dumps(res), flush=True)
