"""Room geometry of a scene export: the triangles, their materials / sidedness, sources and receivers.

Host-side mirror of the reference's `RoomGeo` (python/common/room_geo.py:27-187) and `tris_precompute`
(python/common/tris_precompute.py:21-123), with the same attribute names so that the voxelizer and `sim_setup`
read like the reference's.  The per-triangle quantities are produced with the same numpy operations in the same
order (they feed exact comparisons in the voxelizer, so the last bit matters).  Drawing is out of scope.
"""
import json

import numpy as np

_EPS = np.finfo(np.float64).eps  # python/common/myfuncs.py:29


def rotate_az_el_deg(az_d, el_d):
    """R = Raz @ Rel, Rel about the negative y axis, Raz about z (python/common/myfuncs.py:49-83)."""
    thy, thz = np.deg2rad(-el_d), np.deg2rad(az_d)
    Ry = np.array([[np.cos(thy), 0, np.sin(thy)], [0, 1, 0], [-np.sin(thy), 0, np.cos(thy)]])
    Rz = np.array([[np.cos(thz), -np.sin(thz), 0], [np.sin(thz), np.cos(thz), 0], [0, 0, 1]])
    return Rz @ Ry


def _dotv(a, b):
    return np.sum(a * b, axis=-1)


def _unit(v):
    return (v.T / (np.sqrt(_dotv(v, v)) + _EPS)).T  # myfuncs.py:124-125 (eps in the denominator)


def tris_precompute(pts, tris):
    """Per-triangle vertices, area-scaled and unit normals, outward unit edge normals, centroid, bbox, area."""
    a, b, c = pts[tris[:, 0]], pts[tris[:, 1]], pts[tris[:, 2]]
    ab, bc, ca = b - a, c - b, a - c
    nor = (np.cross(ab, -ca) + np.cross(bc, -ab) + np.cross(ca, -bc)) / 3.0
    pre = {
        "v": np.stack([a, b, c], axis=1),
        "nor": nor,
        "unor": _unit(nor),
        "eab_unor": _unit(np.cross(ab, nor)),
        "ebc_unor": _unit(np.cross(bc, nor)),
        "eca_unor": _unit(np.cross(ca, nor)),
        "cent": (a + b + c) / 3.0,
        "bmin": np.minimum(np.minimum(a, b), c),
        "bmax": np.maximum(np.maximum(a, b), c),
        "area": 0.5 * np.sqrt(_dotv(nor, nor)),
    }
    return pre


def _take(pre, keep):
    return {k: v[keep] for k, v in pre.items()}


class RoomGeo:
    def __init__(self, json_file=None, az_el=(0.0, 0.0), area_eps=1e-6, bmin=None, bmax=None):
        if json_file is None:
            raise ValueError("RoomGeo needs the JSON export of the model")
        self.bmin = np.full(3, np.inf) if bmin is None else np.asarray(bmin, dtype=np.float64)
        self.bmax = np.full(3, -np.inf) if bmax is None else np.asarray(bmax, dtype=np.float64)
        self.area_eps = area_eps
        self.R = rotate_az_el_deg(*az_el)
        self.load_json(json_file)
        self.collapse_tris()
        self.calc_volume()

    def print(self, fstring):
        print(f"--ROOM_GEO: {fstring}")

    def load_json(self, json_filename):
        if str(json_filename).endswith(".gz"):
            import gzip
            with gzip.open(json_filename, "rt") as f:
                data = json.load(f)
        else:
            with open(json_filename) as f:
                data = json.load(f)
        mats = data["mats_hash"]
        names = sorted(mats.keys())  # alphabetical, '_RIGID' (unmarked, index -1) last: room_geo.py:78-86
        Nmat = len(names)
        if "_RIGID" in names:
            names.remove("_RIGID")
            names.append("_RIGID")
            Nmat -= 1
        bmin, bmax = self.bmin, self.bmax
        for m in names:
            mats[m]["pts"] = np.array(mats[m]["pts"], dtype=np.float64) @ self.R
            mats[m]["tris"] = np.array(mats[m]["tris"], dtype=np.int64)
            bmin = np.minimum(bmin, mats[m]["pts"].min(axis=0))
            bmax = np.maximum(bmax, mats[m]["pts"].max(axis=0))
        if not data.get("sources") or not data.get("receivers"):
            raise ValueError("sources and receivers have to be defined in the JSON export")
        self.Sxyz = np.atleast_2d(np.array([s["xyz"] for s in data["sources"]], dtype=np.float64)) @ self.R
        self.Rxyz = np.atleast_2d(np.array([r["xyz"] for r in data["receivers"]], dtype=np.float64)) @ self.R
        for P, what in ((self.Sxyz, "source"), (self.Rxyz, "receiver")):
            if not np.all((P > bmin) & (P < bmax)):
                raise ValueError(f"a {what} lies outside the scene bounds")
        self.mats_dict, self.mat_str, self.Nmat = mats, names, Nmat
        self.colors = [mats[m]["color"] for m in names]
        self.bmin, self.bmax = bmin, bmax

    def collapse_tris(self):
        mats, names, Nmat = self.mats_dict, self.mat_str, self.Nmat
        npts = [mats[m]["pts"].shape[0] for m in names]
        offs = np.concatenate([[0], np.cumsum(npts)[:-1]]).astype(np.int64)
        self.pts = np.concatenate([mats[m]["pts"] for m in names], axis=0)
        tris = np.concatenate([mats[m]["tris"] + o for m, o in zip(names, offs)], axis=0)
        if tris.shape[0] < 4:
            raise ValueError("the export must hold at least four triangles")  # room_geo.py:133
        mat_ind = np.concatenate([np.full(mats[m]["tris"].shape[0], i, dtype=np.int8) for i, m in enumerate(names)])
        mat_ind[mat_ind == Nmat] = -1  # anything on _RIGID
        mat_side = np.concatenate([np.asarray(mats[m]["sides"]) for m in names], axis=0)
        if not np.all(mat_side[mat_ind == -1] == 0):
            raise ValueError("unmarked (rigid) triangles must have sidedness 0")
        pre = tris_precompute(self.pts, tris)
        keep = ~(pre["area"] < self.area_eps)  # degenerate triangles are dropped (room_geo.py:173-181)
        self.print(f"{int((~keep).sum())} degenerate triangles deleted")
        self.tris, self.mat_ind, self.mat_side = tris[keep], mat_ind[keep], mat_side[keep]
        self.tris_pre = _take(pre, keep)
        self.calc_areas()

    def calc_areas(self):
        fac = np.zeros(self.mat_side.shape)
        fac[(self.mat_side == 1) | (self.mat_side == 2)] = 1.0
        fac[self.mat_side == 3] = 2.0  # both sides
        self.mat_area = np.array([np.sum((self.tris_pre["area"] * fac)[self.mat_ind == i]) for i in range(self.Nmat)],
                                 dtype=np.float64)

    def calc_volume(self):
        self.vol = np.sum(_dotv(self.tris_pre["cent"], self.tris_pre["nor"])) / 6.0  # divergence theorem
        self.area = np.sum(self.tris_pre["area"])

    def print_stats(self):
        self.print(f"npts =  {self.pts.shape[0]}")
        self.print(f"ntris = {self.tris.shape[0]}")
        self.print(f"bmin = {self.bmin}")
        self.print(f"bmax = {self.bmax}")
        self.print(f"Lxyz = {self.bmax - self.bmin}")
        self.print(f"room vol = {self.vol}m³")
        self.print(f"room SA = {self.area}m²")
        for i in range(self.Nmat):
            self.print(f"mat {i}: {self.mat_str[i]}, {self.mat_area[i]:.3f}m²")
