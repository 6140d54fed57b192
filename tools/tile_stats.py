#!/usr/bin/env python3
"""How much of a reference configuration could step in temporally blocked pairs?  Builds the scene at full size (sim_setup
on this GPU box) and classifies the tiles of the blocked kernels on the host (numpy): a tile (x chunk of 16 planes x row
tile x row segment) is clean when no boundary node lies within one cell of its core.  usage: tile_stats.py <config> [R]"""
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pffdtd_amd import scenes, sim_data  # noqa: E402
from pffdtd_amd.sim_setup import sim_setup  # noqa: E402

name = sys.argv[1]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
with tempfile.TemporaryDirectory() as d:
    mats = scenes.write_materials(Path(d) / "materials")
    folder = Path(d) / name
    sim_setup(**scenes.setup_kwargs(name, folder, mats, save_folder_gpu=folder, compress=0))
    sd = sim_data.SimData.from_folder(folder, "single", build_mask=False)
Nx, Ny, Nz = sd.Nx, sd.Ny, sd.Nz
ii = sd.bn_ixyz
ix, iy, iz = ii // (Ny * Nz), (ii // Nz) % Ny, ii % Nz
print(f"{name}: grid {Nx}x{Ny}x{Nz}, Nb={sd.Nb} ({sd.Nb / sd.Npts * 100:.2f} % of the cells)")
for lw in (64, 32, 16):
    TC, TR, CH = (lw - 2) * 4, 4 * R * (64 // lw), 16
    x0, y0, z0 = 3, 3, 4
    nx, ny, nz = -(-(Nx - 6) // CH), -(-(Ny - 6) // TR), -(-(Nz - 8) // TC)
    dirty = np.zeros((nx, ny, nz), dtype=bool)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                tx, ty, tz = (ix + dx - x0) // CH, (iy + dy - y0) // TR, (iz + dz - z0) // TC
                ok = (tx >= 0) & (tx < nx) & (ty >= 0) & (ty < ny) & (tz >= 0) & (tz < nz)
                dirty[tx[ok], ty[ok], tz[ok]] = True
    print(f"  lanes/segment {lw}: tiles {CH}x{TR}x{TC}: {dirty.size} tiles, clean {100 * (1 - dirty.mean()):.1f} %")
