#!/usr/bin/env python3
"""The reference's test-script configurations at FULL size (BASELINE configs[1] CTK 894x579x309 7-point, configs[2]
Musikverein 2852x552x850 folded 13-point): the scene is voxelised and GPU-prepared on this box (sim_setup), then several
independent interior-kernel families step it from the same seeded random fields and must leave identical bits in every
cell and at every receiver (the CPU oracle is too slow at these sizes; tests/test_sim_setup.py pins the same geometries
against the oracle at coarse resolution).   usage: tools/config_family_check.py ctk_cart_gpu|mv_fcc_gpu [steps]"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pffdtd_amd import engine, scenes, sim_data  # noqa: E402
from pffdtd_amd.sim_setup import sim_setup  # noqa: E402

name = sys.argv[1]
Nt = int(sys.argv[2]) if len(sys.argv) > 2 else 10
with tempfile.TemporaryDirectory() as d:
    t0 = time.time()
    mats = scenes.write_materials(Path(d) / "materials")
    folder = Path(d) / name
    sim_setup(**scenes.setup_kwargs(name, folder, mats, save_folder_gpu=folder, compress=0))
    sd = sim_data.SimData.from_folder(folder, "single", build_mask=False)
sd.scale_input()
fcc = sd.fcc_flag > 0
n = [sd.Nx, sd.Ny, sd.Nz]
print(f"{name}: grid {n} fcc_flag {sd.fcc_flag}, Nb={sd.Nb} Nbl={sd.Nbl} Nr={sd.Nr}, set up in {time.time()-t0:.1f}s", flush=True)
P = engine.grid_pitch(n[2], 4)
shape = (n[0], n[1] * P)
gen = torch.Generator(device="cuda")
gen.manual_seed(23)
init = [(torch.rand(shape, generator=gen, device="cuda") * 2 - 1) * 1e-3 for _ in range(2)]
g = [torch.empty(shape, dtype=torch.float32, device="cuda") for _ in range(2)]
ref_out, ref_g = None, None
# 13-point: 0 = what the engine picks (barrier-free, ABC in-kernel), 4 = virtual ghosts, 3 = the reference's kernel sequence
# 7-point: 0 = what the engine picks, 25 = lean fused, 4 = barrier-free + virtual ghosts, 3 = the reference's kernel sequence
for v in ((0, 4, 3, 3 + 256) if fcc else (0, 25, 4, 3)):
    for a, b in zip(g, init):
        a.copy_(b)
    sd.u_out[:] = 0
    t0 = time.time()
    eng = engine.HipEngine(sd, air_variant=v, timing=True, ext_u0=g[0].data_ptr(), ext_u1=g[1].data_ptr())
    eng.run(0, Nt)
    eng.sync()
    tm = eng.timing()
    eng.close()
    out = sd.u_out[:, :Nt].copy()
    print(f"variant {v}: {time.time()-t0:.1f}s, air path {tm['air_path']}, blocked launches {tm['tb2_launches']}, "
          f"peak |out| {np.abs(out).max():.3e}", flush=True)
    view = [t.view(n[0], n[1], P)[1:-1, 1:-1, 1:n[2] - 1] for t in g]
    if ref_out is None:
        ref_out, ref_g = out, [t.clone() for t in g]
        assert np.abs(out).max() > 0 and all(bool(torch.isfinite(t).all()) for t in view)
        assert (np.abs(out).max(axis=1) > 0).all(), "a silent receiver"
    else:
        assert np.array_equal(out, ref_out), f"variant {v}: receivers differ, max|d|={np.abs(out-ref_out).max()}"
        for t, r in zip(view, ref_g):
            assert bool(torch.equal(t, r.view(n[0], n[1], P)[1:-1, 1:-1, 1:n[2] - 1])), f"variant {v}: fields differ"
# ... and the arrangement the engine picks by itself for these rooms: its OWN grids, stored with the file's x and z axes exchanged
# (pf_engine_layout).  Same initial field (written and read back in FILE order through pf_engine_set_grid / _get_grid), same
# receivers, same field.
sd.u_out[:] = 0
eng = engine.HipEngine(sd, timing=True)
dims, pitch, exchanged = eng.layout()
for k in range(2):
    eng.set_grid(k, init[k].view(n[0], n[1], P)[:, :, :n[2]].contiguous().cpu().numpy())
t0 = time.time()
eng.run(0, Nt)
eng.sync()
out = sd.u_out[:, :Nt].copy()
print(f"engine-owned grids: stored {dims} pitch {pitch}, axes exchanged: {exchanged}; {time.time()-t0:.1f}s", flush=True)
assert np.array_equal(out, ref_out), f"engine-owned grids: receivers differ, max|d|={np.abs(out-ref_out).max()}"
for k in range(2):
    got = torch.from_numpy(eng.get_grid(k))[1:-1, 1:-1, 1:-1]
    want = ref_g[k].view(n[0], n[1], P)[1:-1, 1:-1, 1:n[2] - 1].cpu()
    assert bool(torch.equal(got, want)), f"engine-owned grids: field {k} differs"
    del got, want
eng.close()
if name in ("ctk_cart_gpu", "mv_fcc_gpu"):
    assert exchanged, "these rooms are expected to be stored with the axes exchanged"
print("config family check OK: all kernel families agree bit for bit on every cell")
