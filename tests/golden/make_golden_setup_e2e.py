#!/usr/bin/env python3
"""End-to-end goldens of the reference's sim_setup() (python/sim_setup.py) on coarse versions of its own test-script
configurations: every dataset it writes to sim_consts.h5, cart_grid.h5, comms_out.h5, sim_mats.h5 and vox_out.h5
(boundary arrays sorted by bn_ixyz).  Build container only; same shims as make_golden_vox.py, with the in-memory
h5py falling back to libhdf5 (pffdtd_amd.h5io) for the material files that exist on disk."""
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
from pffdtd_amd import h5io  # noqa: E402

REF = Path("/root/reference")
np.float = float
np.bool8 = np.bool_
_store = {}


class _DS:
    def __init__(self, a): self.a = np.asarray(a)
    def __getitem__(self, k): return self.a[k] if k != () else (self.a[()] if self.a.ndim == 0 else self.a)


class _File:
    def __init__(self, path, mode="r"):
        self.k = str(path)
        if "w" in mode: _store[self.k] = {}
    def create_dataset(self, name, data=None, **kw): _store[self.k][name] = _DS(np.array(data))
    def __getitem__(self, name):
        if self.k in _store: return _store[self.k][name]
        return _DS(h5io.read(self.k, name))  # a real file (data/materials/*.h5)
    def close(self): pass


h5 = types.ModuleType("h5py"); h5.File = _File; sys.modules["h5py"] = h5
nb = types.ModuleType("numba"); nb.jit = lambda *a, **k: (lambda f: f); nb.prange = range; sys.modules["numba"] = nb
mp_ = types.ModuleType("memory_profiler"); mp_.profile = lambda f: f; sys.modules["memory_profiler"] = mp_
from multiprocessing import shared_memory as _shm  # noqa: E402


class _FakeShm:
    def __init__(self, create=False, size=0): self.buf = memoryview(bytearray(size))
    def close(self): pass
    def unlink(self): pass


_shm.SharedMemory = _FakeShm
sys.path.insert(0, str(REF / "python"))
import voxelizer.vox_scene as _vs  # noqa: E402
_vs.VoxScene.check_adj_full = lambda self: None  # pure-python triple loop over the grid; its asserts cannot fire anyway
from sim_setup import sim_setup  # noqa: E402

CTK = dict(model_json_file=str(REF / "data/models/CTK_Church/model_export.json"), mat_folder=str(REF / "data/materials"),
           mat_files_dict={"AcousticPanel": "ctk_acoustic_panel.h5", "Altar": "ctk_altar.h5", "Carpet": "ctk_carpet.h5",
                           "Ceiling": "ctk_ceiling.h5", "Glass": "ctk_window.h5", "PlushChair": "ctk_chair.h5",
                           "Tile": "ctk_tile.h5", "Walls": "ctk_walls.h5"})
MV = dict(model_json_file=str(REF / "data/models/Musikverein_ConcertHall/model_export.json"), mat_folder=str(REF / "data/materials"),
          mat_files_dict={"Floor": "mv_floor.h5", "Chairs": "mv_chairs.h5", "Plasterboard": "mv_plasterboard.h5",
                          "Window": "mv_window.h5", "Wood": "mv_wood.h5"})
CASES = {
    "ctk_cart": dict(CTK, source_num=1, insig_type="dhann30", diff_source=False, duration=0.02, fcc_flag=False, PPW=6.0, fmax=250.0),
    "ctk_fcc": dict(CTK, source_num=2, insig_type="impulse", diff_source=True, duration=0.02, fcc_flag=True, PPW=6.0, fmax=350.0),
    "mv_fcc": dict(MV, source_num=3, insig_type="impulse", diff_source=True, duration=0.01, fcc_flag=True, PPW=5.0, fmax=500.0),
}
only = sys.argv[1:]
for tag, kw in CASES.items():
    if only and tag not in only:
        continue
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)  # the reference drops its mmap_dat scratch folder into the cwd
        folder = Path(d) / "sim"
        folder.mkdir()
        sim_setup(save_folder=str(folder), compress=0, Nprocs=1, **kw)
        out = {}
        for f in ("sim_consts", "cart_grid", "comms_out", "sim_mats", "vox_out"):
            for name, ds in _store[str(folder / f"{f}.h5")].items():
                out[f"{f}/{name}"] = ds.a
        o = np.argsort(out["vox_out/bn_ixyz"], kind="stable")
        for k in ("bn_ixyz", "adj_bn", "mat_bn", "saf_bn"):
            out[f"vox_out/{k}"] = out[f"vox_out/{k}"][o]
        out["vox_out/adj_bn"] = np.packbits(out["vox_out/adj_bn"], axis=1, bitorder="little")
        np.savez_compressed(HERE / f"setup_e2e_{tag}.npz", **out)
        print(f"{tag}: grid {out['vox_out/Nx']}x{out['vox_out/Ny']}x{out['vox_out/Nz']} Nb={out['vox_out/Nb']} Nt={out['comms_out/Nt']}", flush=True)
