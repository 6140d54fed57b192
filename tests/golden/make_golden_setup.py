#!/usr/bin/env python3
"""Golden vectors of the reference's setup-side math (SimConsts, SimComms) for tests/test_setup_io.py.
The reference modules are imported from /root/reference behind a test-only empty h5py module (this container only)."""
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
np.float = float
sys.modules.setdefault("h5py", types.ModuleType("h5py"))
sys.path.insert(0, "/root/reference/python")
from fdtd.sim_consts import SimConsts  # noqa: E402
from fdtd.sim_comms import SimComms  # noqa: E402

out = {}
for tag, fcc in (("cart", False), ("fcc", True)):
    sc = SimConsts(Tc=20, rh=50, fmax=700.0, PPW=8.0, fcc=fcc)
    for k in ("h", "c", "Ts", "SR", "l", "l2"):
        out[f"{tag}_{k}"] = np.float64(getattr(sc, k))
    N = (20, 18, 16)
    xv, yv, zv = (np.arange(n) * sc.h + o for n, o in zip(N, (-0.3, 0.1, 0.05)))
    cm = SimComms.__new__(SimComms)
    cm.h, cm.Ts, cm.l2, cm.fcc_flag, cm.fcc = sc.h, sc.Ts, sc.l2, int(fcc), fcc
    cm.xv, cm.yv, cm.zv, cm._diff = xv, yv, zv, False
    S = np.array([xv[7] + 0.37 * sc.h, yv[9] - 0.21 * sc.h, zv[6] + 0.5 * sc.h])
    R = np.array([[xv[11] - 0.1 * sc.h, yv[5] + 0.45 * sc.h, zv[9] + 0.3 * sc.h], [xv[5] + 0.6 * sc.h, yv[12] - 0.4 * sc.h, zv[4] + 0.25 * sc.h]])
    cm.prepare_source_pts(S)
    cm.prepare_receiver_pts(R)
    cm.prepare_source_signals(60 * sc.Ts, sig_type="dhann30")
    out[f"{tag}_S"], out[f"{tag}_R"], out[f"{tag}_xv0"] = S, R, np.array([xv[0], yv[0], zv[0]])
    out[f"{tag}_in_alpha"], out[f"{tag}_in_ixyz"] = cm.in_alpha, cm.in_ixyz
    out[f"{tag}_out_alpha"], out[f"{tag}_out_ixyz"] = cm.out_alpha, cm.out_ixyz
    out[f"{tag}_in_sigs"] = cm.in_sigs.copy()
    cm.diff_source()
    out[f"{tag}_in_sigs_diff"] = cm.in_sigs.copy()
np.savez_compressed(HERE / "setup_reference.npz", **out)
print("wrote", HERE / "setup_reference.npz", len(out), "arrays")
