#!/usr/bin/env python3
"""How many distinct 128-byte lines of u1 / u0 does the boundary pass of a scene have to touch?  (Lower bound of its
grid traffic, to judge what a better node order could save.)  usage: tools/boundary_lines.py ctk_cart_gpu"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pffdtd_amd import engine, scenes, setup_io  # noqa: E402
from pffdtd_amd.room_geo import RoomGeo  # noqa: E402
from pffdtd_amd.voxelizer import VoxScene  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ctk_cart_gpu"
cfg = scenes.CONFIGS[name]
rg = RoomGeo(str(scenes.model_path(cfg["model"])))
sc = setup_io.SimConsts(Tc=20, rh=50, fmax=cfg["fmax"], PPW=cfg["PPW"], fcc=cfg["fcc_flag"])
cg = setup_io.CartGrid(h=sc.h, offset=3.5, bmin=rg.bmin, bmax=rg.bmax, fcc=cfg["fcc_flag"])
vs = VoxScene(rg, cg, fcc=cfg["fcc_flag"])
vs.calc_adj()
# engine layout: x = longest axis ... here just use the file order (Nx,Ny,Nz) with a 128-byte pitch
Nx, Ny, Nz = cg.Nx, cg.Ny, cg.Nz
P = engine.grid_pitch(Nz, 4)
ii = vs.bn_ixyz
iz, iy, ix = ii % Nz, (ii // Nz) % Ny, ii // (Nz * Ny)
pad = (ix * Ny + iy) * P + iz
plane = Ny * P
offs = [0, plane, -plane, P, -P, 1, -1]
lines_u1 = np.unique(np.concatenate([(pad + o) // 32 for o in offs]))
lines_u0 = np.unique(pad // 32)
nb = ii.size
print(f"{name}: grid {Nx}x{Ny}x{Nz} pitch {P}, Nb={nb}")
print(f"distinct 128-B lines: u1 {lines_u1.size} ({lines_u1.size*128/1e6:.0f} MB = {lines_u1.size*128/nb:.0f} B/node), "
      f"u0 {lines_u0.size} ({lines_u0.size*128/1e6:.0f} MB read + as much written)")
