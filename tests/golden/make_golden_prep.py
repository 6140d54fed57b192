#!/usr/bin/env python3
"""Goldens of the reference's GPU preparation (python/fdtd/rotate_sim_data.py: rotate_sim_data -> fold_fcc_sim_data ->
sort_sim_data, the order of sim_setup.py:127-133) on small synthetic folders whose axes, boundary-node order and
source / receiver order are deliberately NOT the prepared ones.  Build container only: the reference functions are
imported and run behind an in-memory h5py stand-in (datasets live in a dict); inputs come from
tests/prep_cases.py (deterministic), outputs go to tests/golden/prep_reference_<tag>.npz."""
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE.parent))
import prep_cases  # noqa: E402

_store = {}


class _DS:
    def __init__(self, a): self.a = np.array(a)
    def __getitem__(self, k): return self.a[()] if self.a.ndim == 0 else self.a.copy()
    def __setitem__(self, k, v): self.a = np.array(v, dtype=self.a.dtype).reshape(self.a.shape) if np.shape(v) == self.a.shape else np.array(v, dtype=self.a.dtype)


class _File:
    def __init__(self, path, mode="r"): self.d = _store[str(path)]
    def __getitem__(self, name): return self.d[name]
    def __delitem__(self, name): del self.d[name]
    def create_dataset(self, name, data=None, **kw): self.d[name] = _DS(data)
    def close(self): pass


h5 = types.ModuleType("h5py"); h5.File = _File; sys.modules["h5py"] = h5
nb = types.ModuleType("numba"); nb.jit = lambda *a, **k: (lambda f: f); nb.prange = range; sys.modules["numba"] = nb
np.float = float
np.bool8 = np.bool_
sys.path.insert(0, "/root/reference/python")
from fdtd.rotate_sim_data import fold_fcc_sim_data, rotate_sim_data, sort_sim_data  # noqa: E402

for tag in prep_cases.CASES:
    sim = prep_cases.make(tag)
    d = Path("/virtual") / tag
    for f, dsets in sim.items():
        _store[str(d / f"{f}.h5")] = {k: _DS(v) for k, v in dsets.items()}
    rotate_sim_data(d)
    if int(sim["sim_consts"]["fcc_flag"]) == 1:
        fold_fcc_sim_data(d)
    sort_sim_data(d)
    out = {}
    for f in ("sim_consts", "vox_out", "comms_out"):
        for k, ds in _store[str(d / f"{f}.h5")].items():
            a = ds.a
            out[f"{f}/{k}"] = np.packbits(a.astype(bool), axis=1, bitorder="little") if k == "adj_bn" else a
    np.savez_compressed(HERE / f"prep_reference_{tag}.npz", **out)
    print(tag, "->", [int(out[f"vox_out/{k}"]) for k in ("Nx", "Ny", "Nz")], "fcc_flag", int(out["sim_consts/fcc_flag"]))
