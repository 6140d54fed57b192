#!/usr/bin/env python3
"""Golden vectors of the reference voxelizer (python/voxelizer/vox_scene.py + common/room_geo.py) for
tests/test_voxelizer.py.  Runs in the build container only: the reference modules are imported from
/root/reference behind test-only shims (identity numba.jit, an in-memory h5py, memory_profiler stub).

Scenes: the CTK church export (a data file of the reference's own test scripts, copied to pffdtd_amd/data/models/)
at coarse grid spacings, Cartesian and FCC, plain and rotated; plus the full BASELINE cfg1 resolution (h=0.0915 m).
Stored per case: grid shape, bn_ixyz (sorted), adj_bn packed to uint16, mat_bn, saf_bn, tidx/ndist-independent.
"""
import shutil
import sys
import time
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
DATA = HERE.parent.parent / "pffdtd_amd" / "data"
REF = Path("/root/reference")
np.float = float
np.bool8 = np.bool_

_store = {}


class _DS:
    def __init__(self, a): self.a = np.asarray(a)
    def __getitem__(self, k): return self.a[k]


class _File:
    def __init__(self, path, mode="r"):
        self.k = str(path)
        if "w" in mode: _store[self.k] = {}
    def create_dataset(self, name, data=None, **kw): _store[self.k][name] = _DS(np.array(data))
    def __getitem__(self, name): return _store[self.k][name]
    def close(self): pass


h5 = types.ModuleType("h5py"); h5.File = _File; sys.modules["h5py"] = h5
nb = types.ModuleType("numba")
nb.jit = lambda *a, **k: (lambda f: f)
nb.prange = range
sys.modules["numba"] = nb
mp_ = types.ModuleType("memory_profiler"); mp_.profile = lambda f: f; sys.modules["memory_profiler"] = mp_
from multiprocessing import shared_memory as _shm  # noqa: E402


class _FakeShm:  # the reference closes its SharedMemory while numpy views are alive (BufferError on this Python)
    def __init__(self, create=False, size=0): self.buf = memoryview(bytearray(size))
    def close(self): pass
    def unlink(self): pass


_shm.SharedMemory = _FakeShm
sys.path.insert(0, str(REF / "python"))
from common.room_geo import RoomGeo  # noqa: E402
from voxelizer.cart_grid import CartGrid  # noqa: E402
from voxelizer.vox_grid import VoxGrid  # noqa: E402
from voxelizer.vox_scene import VoxScene  # noqa: E402

MODEL = REF / "data/models/CTK_Church/model_export.json"
dst = DATA / "models" / "CTK_Church_model_export.json"
if not dst.exists():
    shutil.copyfile(MODEL, dst)

CASES = [  # tag, h, fcc, az_el, Nh
    ("ctk_cart_h40", 0.40, False, [0.0, 0.0], None),
    ("ctk_fcc_h40", 0.40, True, [0.0, 0.0], None),
    ("ctk_cart_h25_rot", 0.25, False, [30.0, 10.0], None),
    ("ctk_fcc_h30_rot", 0.30, True, [-15.0, 5.0], 5),
    ("ctk_cart_h0915", 0.0915, False, [0.0, 0.0], None),  # BASELINE cfg1 spacing (test_script_CTK_cart_viz.py)
    # BASELINE cfg2 (test_script_CTK_cart_gpu.py: fmax=1400, PPW=10.5 -> h = c/(fmax*PPW), 894x579x309): digests only
    ("ctk_cart_cfg2_digest", 343.2 / (1400.0 * 10.5), False, [0.0, 0.0], None),
    # Musikverein export (32k triangles, mostly chairs): coarse full arrays + test_script_MV_fcc_viz.py resolution digests
    ("mv_fcc_h20", 0.20, True, [0.0, 0.0], None),
    ("mv_cart_h25", 0.25, False, [0.0, 0.0], None),
    ("mv_fcc_viz_digest", 343.2 / (1000.0 * 5.6), True, [0.0, 0.0], None),
]
MV_MODEL = REF / "data/models/Musikverein_ConcertHall/model_export.json"
OPEN_MODEL = DATA / "models" / "open_scene.json"  # make_open_scene.py: _RIGID triangles, open top, custom bounds
OPEN_BOUNDS = (np.array([-0.4, -0.4, -0.3]), np.array([4.5, 3.7, 3.4]))
CASES += [("open_cart_h10", 0.10, False, [0.0, 0.0], None), ("open_fcc_h12", 0.12, True, [20.0, 0.0], None)]
only = sys.argv[1:]
for tag, h, fcc, az_el, Nh in CASES:
    if only and tag not in only:
        continue
    t0 = time.time()
    if tag.startswith("open_"):
        rg = RoomGeo(str(OPEN_MODEL), az_el=az_el, bmin=OPEN_BOUNDS[0].copy(), bmax=OPEN_BOUNDS[1].copy())
    else:
        rg = RoomGeo(str(MV_MODEL if tag.startswith("mv_") else MODEL), az_el=az_el)
    cg = CartGrid(h=h, offset=3.5, bmin=rg.bmin, bmax=rg.bmax, fcc=fcc)
    vg = VoxGrid(rg, cg, Nh=Nh)
    vg.fill(Nprocs=1)
    vs = VoxScene(rg, cg, vg, fcc=fcc)
    vs.calc_adj(Nprocs=1)
    o = np.argsort(vs.bn_ixyz, kind="stable")
    NN = vs.adj_bn.shape[1]
    bits = (vs.adj_bn[o].astype(np.uint16) << np.arange(NN, dtype=np.uint16)).sum(axis=1).astype(np.uint16)
    out = dict(h=np.float64(h), fcc=np.int8(fcc), az_el=np.array(az_el), Nxyz=np.array(cg.Nxyz, dtype=np.int64),
               xyzmin=cg.xyzmin, bmin=rg.bmin, bmax=rg.bmax, ntris=np.int64(rg.tris.shape[0]), vol=np.float64(rg.vol),
               area=np.float64(rg.area), mat_area=rg.mat_area, mat_str=np.array(rg.mat_str),
               bn_ixyz=vs.bn_ixyz[o], adj_bits=bits, mat_bn=vs.mat_bn[o], saf_bn=vs.saf_bn[o])
    if tag.endswith("_digest"):  # too large to commit: keep counts and SHA-256 of the sorted arrays
        import hashlib
        for k in ("bn_ixyz", "adj_bits", "mat_bn", "saf_bn"):
            out[k + "_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(out.pop(k)).tobytes()).hexdigest())
        out["Nb"] = np.int64(vs.bn_ixyz.size)
        out["mat_counts"] = np.bincount(vs.mat_bn.astype(np.int64) + 1, minlength=rg.Nmat + 1)
        out["saf_sum"] = np.float64(np.sum(vs.saf_bn[o]))
    np.savez_compressed(HERE / f"vox_{tag}.npz", **out)
    print(f"{tag}: grid {cg.Nxyz} Nb={vs.bn_ixyz.size} in {time.time()-t0:.1f}s", flush=True)
