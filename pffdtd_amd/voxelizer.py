"""'Voxelizer': which legs of the FDTD mesh are cut by the room's surfaces -> boundary nodes, adjacency, materials,
surface-area corrections (vox_out.h5).

Host-side mirror of the reference's `VoxScene` (python/voxelizer/vox_scene.py:63-528) with the same attribute and
method names (`calc_adj`, `check_adj_full`, `save`, `bn_ixyz`, `adj_bn`, `mat_bn`, `saf_bn`).  The ray-triangle work
runs on the MI355X through include/pffdtd_vox.h; the reference's `VoxGrid` hierarchy (vox_grid.py, vox_grid_base.py)
has no counterpart here -- candidate triangles are binned on the device.  No CPU fallback: without the HIP library
and a GPU `calc_adj` raises.

    python -m pffdtd_amd.voxelizer --json model_export.json --h 0.1 [--fcc] [--save_folder out] [--check_full]
"""
import argparse
import ctypes
import time
from pathlib import Path

import numpy as np

from . import engine, h5io
from .room_geo import RoomGeo, _dotv, _unit

R_EPS = 1e-6  # relative (to the leg length) slack for near hits: vox_scene.py:60
TRI_DOUBLES = 30

VV_CART = np.array([[1.0, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])
VV_FCC = np.array([[+1.0, +1, 0], [-1, -1, 0], [0, +1, +1], [0, -1, -1], [+1, 0, +1], [-1, 0, -1],
                   [+1, -1, 0], [-1, +1, 0], [0, +1, -1], [0, -1, +1], [+1, 0, -1], [-1, 0, +1]])


class PfVoxDesc(ctypes.Structure):
    _fields_ = [("Nx", ctypes.c_int64), ("Ny", ctypes.c_int64), ("Nz", ctypes.c_int64),
                ("xv", ctypes.c_void_p), ("yv", ctypes.c_void_p), ("zv", ctypes.c_void_p),
                ("NN", ctypes.c_int32), ("fcc", ctypes.c_int32),
                ("vvh", ctypes.c_void_p), ("ray_un", ctypes.c_void_p),
                ("h", ctypes.c_double), ("hf", ctypes.c_double), ("hfe", ctypes.c_double), ("hf1", ctypes.c_double),
                ("nb_eps", ctypes.c_double), ("d_eps", ctypes.c_double), ("cp_eps", ctypes.c_double),
                ("Ntris", ctypes.c_int64), ("tris", ctypes.c_void_p),
                ("device", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class PfVoxStats(ctypes.Structure):
    _fields_ = [("ms_bin", ctypes.c_double), ("ms_vox", ctypes.c_double), ("ms_total", ctypes.c_double),
                ("ncells", ctypes.c_int64), ("ncells_nonempty", ctypes.c_int64), ("npairs", ctypes.c_int64),
                ("npoints_tested", ctypes.c_int64)]


EXPORTS = ["pf_vox_run", "pf_vox_count", "pf_vox_fetch", "pf_vox_get_stats", "pf_vox_free"]


def _lib():
    L = engine.lib()
    if not getattr(L, "_vox_ready", False):
        L.pf_vox_run.restype = ctypes.c_void_p
        L.pf_vox_run.argtypes = [ctypes.POINTER(PfVoxDesc)]
        L.pf_vox_count.restype = ctypes.c_int64
        L.pf_vox_count.argtypes = [ctypes.c_void_p]
        L.pf_vox_fetch.restype = ctypes.c_int
        L.pf_vox_fetch.argtypes = [ctypes.c_void_p] * 5
        L.pf_vox_get_stats.restype = ctypes.c_int
        L.pf_vox_get_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(PfVoxStats)]
        L.pf_vox_free.restype = None
        L.pf_vox_free.argtypes = [ctypes.c_void_p]
        L._vox_ready = True
    return L


def pack_triangles(pre, hfe):
    """[Ntris, 30] records for pf_vox_desc.tris; the mid-edge points and the padded box are formed with the
    expressions the reference evaluates per ray (tri_ray_intersection.py:100-102, vox_scene.py:170-171)."""
    a, b, c = pre["v"][:, 0, :], pre["v"][:, 1, :], pre["v"][:, 2, :]
    rec = np.concatenate([pre["cent"], pre["unor"], 0.5 * (a + b), 0.5 * (b + c), 0.5 * (c + a),
                          pre["eab_unor"], pre["ebc_unor"], pre["eca_unor"], pre["bmin"] - hfe, pre["bmax"] + hfe], axis=1)
    assert rec.shape[1] == TRI_DOUBLES
    return np.ascontiguousarray(rec, dtype=np.float64)


def cut_legs(xv, yv, zv, h, fcc, tris_pre, device=0):
    """Device pass.  -> bn_ixyz (ascending), adj_bn bool[Nb,NN], tidx_bn int32[Nb], ndist_bn f64[Nb], stats dict."""
    L = _lib()
    VV = VV_FCC if fcc else VV_CART
    NN = VV.shape[0]
    hf = h * np.sqrt(2.0) if fcc else h
    uvv = VV / np.sqrt(2.0) if fcc else VV
    vvh = np.ascontiguousarray(h * VV)
    ray_un = np.ascontiguousarray(np.stack([_unit(uvv[k] * np.ones((1, 3)))[0] for k in range(NN)]))
    hfe = hf * (1 + R_EPS)
    tris = pack_triangles(tris_pre, hfe)
    xv, yv, zv = (np.ascontiguousarray(v, dtype=np.float64) for v in (xv, yv, zv))
    d = PfVoxDesc()
    d.Nx, d.Ny, d.Nz = xv.size, yv.size, zv.size
    d.xv, d.yv, d.zv = xv.ctypes.data, yv.ctypes.data, zv.ctypes.data
    d.NN, d.fcc = NN, int(bool(fcc))
    d.vvh, d.ray_un = vvh.ctypes.data, ray_un.ctypes.data
    d.h, d.hf, d.hfe, d.hf1 = h, hf, hfe, (1 + R_EPS) * hf
    d.nb_eps, d.d_eps, d.cp_eps = R_EPS * hf, abs(1.0e-3 * h), 1e-6
    d.Ntris, d.tris = tris.shape[0], tris.ctypes.data
    d.device = device
    job = L.pf_vox_run(ctypes.byref(d))
    if not job:
        raise engine.PfError(L.pf_last_error().decode())
    try:
        n = L.pf_vox_count(job)
        idx = np.empty(n, dtype=np.int64)
        cut = np.empty(n, dtype=np.uint16)
        tidx = np.empty(n, dtype=np.int32)
        nd = np.empty(n, dtype=np.float64)
        if L.pf_vox_fetch(job, idx.ctypes.data, cut.ctypes.data, tidx.ctypes.data, nd.ctypes.data):
            raise engine.PfError(L.pf_last_error().decode())
        st = PfVoxStats()
        L.pf_vox_get_stats(job, ctypes.byref(st))
    finally:
        L.pf_vox_free(job)
    o = np.argsort(idx, kind="stable")
    idx, cut, tidx, nd = idx[o], cut[o], tidx[o], nd[o]
    adj = ((cut[:, None] >> np.arange(NN, dtype=np.uint16)[None, :]) & 1) == 0
    stats = {k: getattr(st, k) for k, _ in PfVoxStats._fields_}
    return idx, adj, tidx, nd, stats


class VoxScene:
    def __init__(self, room_geo=None, cart_grid=None, vox_grid=None, fcc=False, device=0):
        self.room_geo, self.cart_grid, self.fcc, self.device = room_geo, cart_grid, bool(fcc), device
        h = cart_grid.h
        self.NN = 12 if fcc else 6
        self.hf = h * np.sqrt(2.0) if fcc else h
        self.face_area = h * h / np.sqrt(2.0) if fcc else h * h
        self.VV = VV_FCC if fcc else VV_CART
        self.uvv = self.VV / np.sqrt(2.0) if fcc else self.VV
        self.vvh = h * self.VV

    def print(self, fstring):
        print(f"--VOX_SCENE: {fstring}")

    def calc_adj(self, Nprocs=None):
        cg, rg = self.cart_grid, self.room_geo
        t0 = time.perf_counter()
        bn_ixyz, adj_bn, tidx_bn, ndist_bn, self.stats = cut_legs(cg.xv, cg.yv, cg.zv, cg.h, self.fcc, rg.tris_pre,
                                                                   device=self.device)
        self.print(f"Nbt={bn_ixyz.size}  ({self.stats['ncells_nonempty']} of {self.stats['ncells']} cells, "
                   f"{self.stats['npairs']} cell-triangle pairs; device {self.stats['ms_bin'] + self.stats['ms_vox']:.2f} ms)")
        # materials: the nearest triangle's, unless the point sits on the wrong side of a one-sided triangle or on
        # a surface (vox_scene.py:393-404)
        Ny, Nz = cg.Ny, cg.Nz
        iz = bn_ixyz % Nz
        iy = (bn_ixyz - iz) // Nz % Ny
        ix = ((bn_ixyz - iz) // Nz - iy) // Ny
        pre = rg.tris_pre
        U, C = pre["unor"][tidx_bn], pre["cent"][tidx_bn]  # nearest triangle's unit normal and centroid, gathered once

        def dot3(a0, a1, a2, b):  # np.sum(a*b, axis=-1) over three elements, as (p0+p1)+p2 column-wise (much faster)
            return (a0 * b[:, 0] + a1 * b[:, 1]) + a2 * b[:, 2]
        dv = dot3(cg.xv[ix] - C[:, 0], cg.yv[iy] - C[:, 1], cg.zv[iz] - C[:, 2], U)
        side = rg.mat_side[tidx_bn]
        mat_bn = rg.mat_ind[tidx_bn].copy()
        mat_bn[(dv > 0) & (side == 1)] = -1
        mat_bn[(dv < 0) & (side == 2)] = -1
        mat_bn[np.all(~adj_bn, axis=-1)] = -1
        self.print(f"Npts = {cg.Npts}, Nbl = {np.sum(mat_bn > -1)}")
        # surface-area factors: |leg direction . unit normal| per cut leg pair (vox_scene.py:412-419)
        saf_bn = np.zeros(bn_ixyz.size, dtype=np.float64)
        for j in range(0, self.NN, 2):
            saf = np.abs(dot3(self.uvv[j][0], self.uvv[j][1], self.uvv[j][2], U))
            saf_bn += (~adj_bn[:, j] + ~adj_bn[:, j + 1]) * saf
        # per-material surface totals, for the printout only (the reference uses np.add.at; bincount is the same sums)
        mi = np.where(mat_bn < 0, rg.Nmat, mat_bn).astype(np.int64)  # -1 (rigid) goes to the end
        sa = np.bincount(mi, weights=self.face_area * saf_bn, minlength=rg.Nmat + 1)
        sa0 = np.bincount(mi, weights=self.face_area * np.sum(~adj_bn, axis=-1), minlength=rg.Nmat + 1)
        for i in range(rg.Nmat):
            if rg.mat_area[i] > 0:
                self.print(f"mat: {rg.mat_str[i]}, original: {(sa0[i] / rg.mat_area[i] - 1) * 100.:.3f}% over, "
                           f"corrected: {(sa[i] / rg.mat_area[i] - 1) * 100:.3f}% over")
        self.bn_ixyz, self.adj_bn, self.mat_bn, self.saf_bn = bn_ixyz, adj_bn, mat_bn, saf_bn
        self.tidx_bn, self.ndist_bn = tidx_bn, ndist_bn
        self.print(f"calc_adj total: {time.perf_counter() - t0:.3f} s")

    def check_adj_full(self):
        """Every cut leg must be cut from both ends (the reference's intent at vox_scene.py:497-528; its numba
        asserts test `~(a ^ b)` on integers, which is never zero, so they cannot fire).  Returns the number of
        one-sided legs; the outermost grid shell is skipped like there."""
        cg = self.cart_grid
        Nx, Ny, Nz = cg.Nx, cg.Ny, cg.Nz
        bn, adj = self.bn_ixyz, self.adj_bn
        iv = np.rint(self.VV).astype(np.int64)
        bad = 0
        iz = bn % Nz
        iy = (bn // Nz) % Ny
        ix = bn // (Nz * Ny)
        for k in range(self.NN):
            ko = k + 1 if k % 2 == 0 else k - 1  # legs come in opposite pairs
            jx, jy, jz = ix + iv[k, 0], iy + iv[k, 1], iz + iv[k, 2]
            inside = (jx >= 1) & (jx <= Nx - 2) & (jy >= 1) & (jy <= Ny - 2) & (jz >= 1) & (jz <= Nz - 2)
            nbr = (jx * Ny + jy) * Nz + jz
            pos = np.searchsorted(bn, nbr)
            pos[pos >= bn.size] = 0
            is_bn = (bn[pos] == nbr) & inside
            nbr_adj = np.where(is_bn, adj[pos, ko], True)  # not a boundary node: all its legs are intact
            bad += int(np.sum((adj[:, k] != nbr_adj) & inside))
        self.print(f"check_adj_full: {bad} one-sided legs")
        return bad

    def save(self, save_folder, compress=None):
        cg = self.cart_grid
        save_folder = Path(save_folder)
        save_folder.mkdir(parents=True, exist_ok=True)
        f = save_folder / "vox_out.h5"
        data = {"bn_ixyz": self.bn_ixyz, "adj_bn": self.adj_bn, "mat_bn": self.mat_bn.astype(np.int8),
                "saf_bn": self.saf_bn, "xv": cg.xv, "yv": cg.yv, "zv": cg.zv, "h": np.float64(cg.h),
                "Nx": np.int64(cg.Nx), "Ny": np.int64(cg.Ny), "Nz": np.int64(cg.Nz), "Nb": np.int64(self.bn_ixyz.size)}
        for i, (k, v) in enumerate(data.items()):  # names / dtypes of vox_scene.py:476-488
            h5io.write(f, k, v, append=i > 0, gzip=int(compress or 0) if np.ndim(v) else 0)
        self.print(f"saved {save_folder / 'vox_out.h5'}")


def main():
    from .setup_io import CartGrid
    p = argparse.ArgumentParser()
    p.add_argument("--json", type=str, required=True, help="json file to import")
    p.add_argument("--h", type=float, required=True, help="grid spacing")
    p.add_argument("--fcc", action="store_true")
    p.add_argument("--offset", type=float, default=3.0)
    p.add_argument("--area_eps", type=float, default=1.0e-10)
    p.add_argument("--az_el", nargs=2, type=float, default=[0.0, 0.0])
    p.add_argument("--check_full", action="store_true")
    p.add_argument("--save_folder", type=str, default=None)
    p.add_argument("--gpu", type=int, default=0)
    a = p.parse_args()
    rg = RoomGeo(a.json, az_el=a.az_el, area_eps=a.area_eps)
    rg.print_stats()
    cg = CartGrid(a.h, a.offset, rg.bmin, rg.bmax, fcc=a.fcc)
    vs = VoxScene(rg, cg, fcc=a.fcc, device=a.gpu)
    vs.calc_adj()
    if a.check_full:
        vs.check_adj_full()
    if a.save_folder:
        vs.save(a.save_folder)


if __name__ == "__main__":
    main()
