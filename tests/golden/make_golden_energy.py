#!/usr/bin/env python3
"""Golden vectors of the reference PYTHON engine (python/fdtd/sim_fdtd.py: fp64, energy diagnostic) for a few
tiny scenes.  The reference is imported from /root/reference (this container only) behind test-only shims:
numba is absent (jit -> identity, prange -> range: the kernels run as plain Python loops), h5py is absent
(load_h5_data is bypassed by assigning the attributes it would set), and np.float was removed from numpy.
Nothing of the reference is copied: the fixtures hold inputs' digest + outputs (u_out, H_tot, E_lost, E_in).
"""
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import cases  # noqa: E402
from make_golden import digest  # noqa: E402

np.float = float
np.bool8 = np.bool_
nb = types.ModuleType("numba")
nb.jit = lambda *a, **k: (lambda f: f)
nb.prange = range
nb.set_num_threads = lambda n: None
sys.modules["numba"] = nb
sys.modules["h5py"] = types.ModuleType("h5py")
for m in ("tqdm",):
    try:
        __import__(m)
    except ImportError:
        t = types.ModuleType(m)
        t.tqdm = lambda *a, **k: None
        sys.modules[m] = t
sys.path.insert(0, "/root/reference/python")
from fdtd.sim_fdtd import SimEngine, MMb  # noqa: E402

from energy_cases import ENERGY_CASES  # noqa: E402


def run_reference(sim):
    k, v, c, m = sim["sim_consts"], sim["vox_out"], sim["comms_out"], sim["sim_mats"]
    e = SimEngine.__new__(SimEngine)
    e.energy_on = True
    e.adj_bn, e.bn_ixyz = np.asarray(v["adj_bn"], dtype=bool), np.asarray(v["bn_ixyz"])
    e.Nx, e.Ny, e.Nz = int(v["Nx"]), int(v["Ny"]), int(v["Nz"])
    mat_bn, saf_bn = np.asarray(v["mat_bn"]), np.asarray(v["saf_bn"])
    ii = mat_bn > -1
    e.saf_bnl, e.mat_bnl, e.bnl_ixyz = saf_bn[ii], mat_bn[ii], e.bn_ixyz[ii]
    e.in_ixyz, e.out_ixyz, e.out_reorder = c["in_ixyz"], c["out_ixyz"], c["out_reorder"]
    e.in_sigs, e.Ns, e.Nr, e.Nt = np.array(c["in_sigs"]), int(c["Ns"]), int(c["Nr"]), int(c["Nt"])
    e.c, e.h, e.Ts, e.l, e.l2 = float(k["c"]), float(k["h"]), float(k["Ts"]), float(k["l"]), float(k["l2"])
    e.fcc_flag = int(k["fcc_flag"])
    e.fcc = e.fcc_flag > 0
    e.ssaf_bnl = e.saf_bnl * 0.5 / np.sqrt(2.0) if e.fcc else e.saf_bnl
    Nmat = int(m["Nmat"])
    DEF = np.zeros((Nmat, MMb, 3))
    for i in range(Nmat):
        d = np.asarray(m[f"mat_{i:02d}_DEF"])
        DEF[i, :d.shape[0]] = d
    e.DEF, e.Nm, e.Mb = DEF, Nmat, np.asarray(m["Mb"])
    e._load_abc()
    e.setup_mask()
    e.allocate_mem()
    e.set_coeffs()
    e.checks()
    for n in range(e.Nt):
        e.run_steps(n, 1)  # the energy diagnostic is only valid with nsteps=1
    return e.u_out, e.H_tot, e.E_lost, e.E_in


def main():
    from pffdtd_amd import synth
    for name, kw in ENERGY_CASES.items():
        sim = synth.shoebox(**kw)
        u_out, H, El, Ei = run_reference(sim)
        bal = (H + El[:-1] - Ei[:-1])
        scale = np.maximum(np.abs(H + El[:-1]), np.abs(Ei[:-1])).max()
        out = HERE / f"energy_{name}.npz"
        np.savez_compressed(out, u_out=u_out, H_tot=H, E_lost=El, E_in=Ei, digest=np.array(digest(sim)))
        print(f"{out.name}: peak u {np.abs(u_out).max():.4e}  max|balance|/scale {np.abs(bal).max()/scale:.3e}")


if __name__ == "__main__":
    main()
