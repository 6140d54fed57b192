"""Synthetic shoebox scenes in the reference's on-disk schema.

The reference's voxelizer (python/voxelizer/*) is out of scope; this generator produces the same
four-file contract (sim_consts.h5, vox_out.h5, comms_out.h5, sim_mats.h5) for a box-shaped room so the
engine, the oracle and the compiled reference can all be driven from identical inputs.  Everything is
deterministic (no RNG).  Also holds the fold / sort transforms the multi-GPU engine expects
(semantics of python/fdtd/rotate_sim_data.py:132-262, re-implemented on in-memory dicts).

Geometry: a wall surface sits between node layers `wall-1` and `wall` (and between N-1-wall and
N-wall) on every axis; nodes inside are air, the inside layer is lossy (material >= 0), the outside
layer is rigid (material -1) -- so boundary nodes never touch the ghost/ABC shell, like
CartGrid(offset=3.5) guarantees in the reference (python/sim_setup.py:91).
"""
from pathlib import Path

import numpy as np

from . import h5io

CART_OFFS = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.int64)
# neighbour order of the 12-point stencil = adjacency bit order (cpu_engine.h:273-284)
FCC_OFFS = np.array([[+1, +1, 0], [-1, -1, 0], [0, +1, +1], [0, -1, -1], [+1, 0, +1], [-1, 0, -1],
                     [+1, -1, 0], [-1, +1, 0], [0, +1, -1], [0, -1, +1], [+1, 0, -1], [-1, 0, +1]], dtype=np.int64)


def make_materials(Nm, Mb):
    """Deterministic positive (D,E,F) triplets of the magnitude found in data/materials/*.h5.

    Branch m of material k is a series RLC resonant at an octave-band centre with a k-dependent damping.
    """
    Mb = np.broadcast_to(np.asarray(Mb, dtype=np.int8), (Nm,)).copy()
    DEF = []
    for k in range(Nm):
        M = int(Mb[k])
        fc = 16.0 * 2.0 ** np.arange(M)
        w0 = 2 * np.pi * fc
        E = 2.0 + 30.0 * (1 + k) / (1.0 + 0.35 * np.arange(M))
        bw = 0.8 + 0.1 * k
        D = E / (w0 * bw)
        F = D * w0 * w0
        DEF.append(np.stack([D, E, F], axis=1).astype(np.float64))
    return Mb, DEF


def _shell_candidates(Nx, Ny, Nz, w):
    """Sorted linear indices of all nodes that can be boundary nodes: the 2-node-thick skin of the box."""
    lo, hi = w - 1, np.array([Nx, Ny, Nz]) - w  # expanded box [lo, hi] inclusive
    ys = np.arange(lo, hi[1] + 1)
    zs = np.arange(lo, hi[2] + 1)
    yy, zz = np.meshgrid(ys, zs, indexing="ij")
    skin2 = (yy <= w) | (yy >= Ny - 1 - w) | (zz <= w) | (zz >= Nz - 1 - w)
    full2 = (yy * Nz + zz).ravel()
    ring2 = (yy * Nz + zz)[skin2].ravel()
    out = []
    for ix in range(lo, hi[0] + 1):
        plane = full2 if (ix <= w or ix >= Nx - 1 - w) else ring2
        out.append(ix * Ny * Nz + plane)
    return np.concatenate(out)


def _ind2sub(ii, Ny, Nz):
    iz = ii % Nz
    iy = (ii // Nz) % Ny
    ix = ii // (Nz * Ny)
    return ix, iy, iz


def shoebox(Nx, Ny, Nz, Nt, fcc=False, wall=3, Nm=1, Mb=2, lossy=True, rigid_every=0,
            src=None, rcv=None, sig="impulse", diff=True, h=0.05, c=343.2, box=True, blocks=()):
    """Return the in-memory file contract for a shoebox room: dict of {file: {dataset: array}}.

    fcc=True gives fcc_flag 1 (checkerboard subgrid, python engine / C-CPU form); use fold_fcc() for flag 2.
    box=False gives free space (no boundary nodes: only the ABC shell terminates the grid).  Note that the
    box is closed: a source inside never reaches the ghost/ABC shell, a source outside (src=...) never enters.
    blocks = solid cuboids (x0, x1, y0, y1, z0, z1), inclusive cell ranges, standing inside the room (interior geometry:
    pillars, balconies): their cells are outside the air domain and their surfaces get boundary nodes like the walls.
    """
    assert min(Nx, Ny, Nz) >= 2 * wall + 3, "grid too small for the wall offset"
    if fcc:
        assert Nx % 2 == 0 and Ny % 2 == 0 and Nz % 2 == 0  # sim_fdtd.py:103-106
    offs = FCC_OFFS if fcc else CART_OFFS
    NN = offs.shape[0]
    w = wall
    dims = np.array([Nx, Ny, Nz])

    cand = _shell_candidates(Nx, Ny, Nz, w) if box else np.zeros((0,), dtype=np.int64)
    if len(blocks):  # every cell within one cell of a block (inside or outside it)
        extra = []
        for (x0, x1, y0, y1, z0, z1) in blocks:
            assert w < x0 <= x1 < Nx - 1 - w and w < y0 <= y1 < Ny - 1 - w and w < z0 <= z1 < Nz - 1 - w, "blocks stand inside the room"
            gx, gy, gz = np.meshgrid(np.arange(x0 - 1, x1 + 2), np.arange(y0 - 1, y1 + 2), np.arange(z0 - 1, z1 + 2), indexing="ij")
            deep = ((gx > x0) & (gx < x1) & (gy > y0) & (gy < y1) & (gz > z0) & (gz < z1))  # not next to the surface
            extra.append(((gx * Ny + gy) * Nz + gz)[~deep].ravel())
        cand = np.unique(np.concatenate([cand] + extra)).astype(np.int64)
    cx, cy, cz = _ind2sub(cand, Ny, Nz)
    if fcc:
        keep = ((cx + cy + cz) % 2) == 0
        cand, cx, cy, cz = cand[keep], cx[keep], cy[keep], cz[keep]

    def inside(x, y, z):
        ok = ((x >= w) & (x <= Nx - 1 - w) & (y >= w) & (y <= Ny - 1 - w) & (z >= w) & (z <= Nz - 1 - w))
        for (x0, x1, y0, y1, z0, z1) in blocks:
            ok = ok & ~((x >= x0) & (x <= x1) & (y >= y0) & (y <= y1) & (z >= z0) & (z <= z1))
        return ok

    ins = inside(cx, cy, cz)
    adj = np.empty((cand.size, NN), dtype=np.bool_)
    for j in range(NN):
        adj[:, j] = inside(cx + offs[j, 0], cy + offs[j, 1], cz + offs[j, 2]) == ins
    isbn = ~np.all(adj, axis=1)
    bn_ixyz = cand[isbn].astype(np.int64)
    adj_bn = adj[isbn]
    ins = ins[isbn]
    bx, by, bz = cx[isbn], cy[isbn], cz[isbn]
    saf_bn = (NN - adj_bn.sum(axis=1)).astype(np.float64)
    mat_bn = np.full(bn_ixyz.shape, -1, dtype=np.int8)
    if lossy:
        mat_bn[ins] = ((bx + 2 * by + 3 * bz)[ins] % Nm).astype(np.int8)
        if rigid_every:
            k = np.flatnonzero(ins)[::rigid_every]
            mat_bn[k] = -1

    # constants (python/fdtd/sim_consts.py:29-52)
    l = (1.0 if fcc else np.sqrt(1.0 / 3.0)) * 0.999
    l2 = l * l
    Ts = h / c * l

    # sources / receivers: trilinear corner nodes of a grid cell (python/fdtd/sim_comms.py:176-231)
    def corners(p, frac):
        p = np.asarray(p, dtype=np.int64)
        nodes, alpha = [], []
        for dx in (0, 1):
            for dy in (0, 1):
                for dz in (0, 1):
                    q = p + np.array([dx, dy, dz])
                    if fcc and (q.sum() % 2) != 0:
                        continue
                    a = ((frac[0] if dx else 1 - frac[0]) * (frac[1] if dy else 1 - frac[1]) *
                         (frac[2] if dz else 1 - frac[2]))
                    nodes.append(q[0] * Ny * Nz + q[1] * Nz + q[2])
                    alpha.append(a)
        alpha = np.array(alpha)
        return np.array(nodes, dtype=np.int64), alpha / alpha.sum()

    if src is None:
        src = (dims // 2).tolist()
    if rcv is None:
        rcv = [[w + 2 + (Nx - 2 * w - 5) // 4, w + 2 + (Ny - 2 * w - 5) // 3, w + 2 + (Nz - 2 * w - 5) // 5],
               [Nx - w - 4 - (Nx - 2 * w - 5) // 5, Ny - w - 4 - (Ny - 2 * w - 5) // 4, w + 2]]
    in_ixyz, in_alpha = corners(src, (0.3, 0.6, 0.2))
    outs = [corners(r, (0.25, 0.5, 0.75)) for r in rcv]
    out_ixyz = np.concatenate([o[0] for o in outs])
    out_alpha = np.stack([o[1] for o in outs])
    bnset = set(bn_ixyz.tolist()) if bn_ixyz.size < 5_000_000 else None
    if bnset is not None:  # sim_comms.py:233-249: sources/receivers never sit on boundary nodes
        assert not (set(in_ixyz.tolist()) & bnset) and not (set(out_ixyz.tolist()) & bnset)

    s = np.zeros(Nt)
    if sig == "impulse":
        s[0] = 1.0
    elif sig == "dhann30":
        n = np.arange(min(30, Nt))
        s[:n.size] = np.cos(np.pi * n / 30) * np.sin(np.pi * n / 30)
    elif sig == "hann10":
        n = np.arange(min(10, Nt))
        s[:n.size] = 0.5 * (1.0 - np.cos(2 * np.pi * n / 10))
    else:
        raise ValueError(f"unknown signal {sig!r}")
    in_sigs = in_alpha[:, None] * s[None, :]
    in_sigs *= (0.5 * l2 / h) if fcc else (l2 / h)          # sim_comms.py:95-104
    if diff:                                                # bilinear differentiator, sim_comms.py:106-119
        y = np.zeros_like(in_sigs)
        prev_x = np.zeros(in_sigs.shape[0])
        prev_y = np.zeros(in_sigs.shape[0])
        for n in range(Nt):
            y[:, n] = (2.0 / Ts) * (in_sigs[:, n] - prev_x) - prev_y
            prev_x, prev_y = in_sigs[:, n], y[:, n]
        in_sigs = y

    Mb_arr, DEF = make_materials(Nm, Mb)
    sim = {
        "sim_consts": {"c": np.float64(c), "h": np.float64(h), "Ts": np.float64(Ts), "SR": np.float64(1 / Ts),
                       "l": np.float64(l), "l2": np.float64(l2), "fcc_flag": np.int8(1 if fcc else 0),
                       "Tc": np.float64(20.0), "rh": np.float64(50.0)},
        "vox_out": {"Nx": np.int64(Nx), "Ny": np.int64(Ny), "Nz": np.int64(Nz), "Nb": np.int64(bn_ixyz.size),
                    "bn_ixyz": bn_ixyz, "adj_bn": adj_bn, "mat_bn": mat_bn, "saf_bn": saf_bn,
                    "xv": np.arange(Nx) * h, "yv": np.arange(Ny) * h, "zv": np.arange(Nz) * h},
        "comms_out": {"Nt": np.int64(Nt), "Ns": np.int64(in_ixyz.size), "Nr": np.int64(out_ixyz.size),
                      "diff": np.int8(1 if diff else 0), "in_ixyz": in_ixyz, "out_ixyz": out_ixyz,
                      "out_reorder": np.arange(out_ixyz.size, dtype=np.int64), "in_sigs": in_sigs,
                      "out_alpha": out_alpha},
        "sim_mats": {"Nmat": np.int8(Nm), "Mb": Mb_arr.astype(np.int8),
                     **{f"mat_{k:02d}_DEF": DEF[k] for k in range(Nm)}},
    }
    return sim


def rotate_sim(sim, tr=None):
    """In-place equivalent of rotate_sim_data (python/fdtd/rotate_sim_data.py:30-130): permute the axes so that
    Nx >= Ny >= Nz (the slab axis is the longest, the plane to exchange the smallest), re-basing every index list and
    permuting the adjacency columns to the new neighbour order.  tr = new-axis -> old-axis permutation."""
    v, c = sim["vox_out"], sim["comms_out"]
    dims = np.array([int(v["Nx"]), int(v["Ny"]), int(v["Nz"])])
    if tr is None:
        tr = np.argsort(dims, kind="stable")[::-1]  # descending
    tr = np.asarray(tr)
    assert sorted(tr.tolist()) == [0, 1, 2]
    if np.array_equal(tr, [0, 1, 2]):
        return sim
    Nx, Ny, Nz = dims.tolist()
    Nt = dims[tr]

    def rot(ii):
        sub = _ind2sub(np.asarray(ii), Ny, Nz)
        new = [sub[t] for t in tr]
        return (new[0] * Nt[1] + new[1]) * Nt[2] + new[2]

    offs = FCC_OFFS if v["adj_bn"].shape[1] == 12 else CART_OFFS
    # column j of the rotated adjacency = the old column whose offset, expressed in the new axes, is offs[j]
    cols = []
    for j in range(offs.shape[0]):
        old = np.zeros(3, dtype=np.int64)
        old[tr] = offs[j]                 # new axis a carries old axis tr[a]
        cols.append(int(np.flatnonzero((offs == old).all(axis=1))[0]))
    v["bn_ixyz"] = rot(v["bn_ixyz"]).astype(np.int64)
    v["adj_bn"] = np.ascontiguousarray(v["adj_bn"][:, cols])
    c["in_ixyz"] = rot(c["in_ixyz"]).astype(np.int64)
    c["out_ixyz"] = rot(c["out_ixyz"]).astype(np.int64)
    v["Nx"], v["Ny"], v["Nz"] = (np.int64(Nt[0]), np.int64(Nt[1]), np.int64(Nt[2]))
    xyz = [v.get("xv"), v.get("yv"), v.get("zv")]
    if all(a is not None for a in xyz):
        v["xv"], v["yv"], v["zv"] = xyz[tr[0]], xyz[tr[1]], xyz[tr[2]]
    return sim


def sort_sim(sim):
    """In-place equivalent of sort_sim_data (python/fdtd/rotate_sim_data.py:132-189)."""
    v, c = sim["vox_out"], sim["comms_out"]
    ii = np.argsort(v["bn_ixyz"], kind="stable")
    for k in ("bn_ixyz", "adj_bn", "mat_bn", "saf_bn"):
        v[k] = v[k][ii]
    ii = np.argsort(c["in_ixyz"], kind="stable")
    c["in_ixyz"] = c["in_ixyz"][ii]
    c["in_sigs"] = c["in_sigs"][ii]
    ii = np.argsort(c["out_ixyz"], kind="stable")
    c["out_ixyz"] = c["out_ixyz"][ii]
    c["out_reorder"] = np.argsort(ii, kind="stable").astype(np.int64)
    return sim


def fold_fcc(sim):
    """In-place equivalent of fold_fcc_sim_data (python/fdtd/rotate_sim_data.py:191-262): flag 1 -> flag 2.

    Rows iy >= Ny/2 are mirrored onto Ny-1-iy of a dense (Nx, Ny/2+1, Nz) grid; for mirrored nodes the
    +y/-y members of the adjacency pairs swap (columns 0<->6, 1<->7, 2<->9, 3<->8).
    """
    v, c, k = sim["vox_out"], sim["comms_out"], sim["sim_consts"]
    assert int(k["fcc_flag"]) == 1
    Nx, Ny, Nz = int(v["Nx"]), int(v["Ny"]), int(v["Nz"])
    assert Ny % 2 == 0
    Nyh = Ny // 2 + 1

    def fold_idx(ii):
        ix, iy, iz = _ind2sub(ii, Ny, Nz)
        up = iy >= Ny // 2
        iyf = np.where(up, Ny - iy - 1, iy)
        return (ix * Nyh + iyf) * Nz + iz, up

    bn, up = fold_idx(v["bn_ixyz"])
    adj = v["adj_bn"].copy()
    for a, b in ((0, 6), (1, 7), (2, 9), (3, 8)):
        ta = adj[up, a].copy()
        adj[up, a] = adj[up, b]
        adj[up, b] = ta
    v["bn_ixyz"] = bn.astype(np.int64)
    v["adj_bn"] = adj
    c["in_ixyz"] = fold_idx(c["in_ixyz"])[0].astype(np.int64)
    c["out_ixyz"] = fold_idx(c["out_ixyz"])[0].astype(np.int64)
    v["Ny"] = np.int64(Nyh)  # (yv stays as it is, like fold_fcc_sim_data, rotate_sim_data.py:191-262: it is only read for plotting)
    k["fcc_flag"] = np.int8(2)
    return sim


def write_folder(sim, data_dir, gzip=0):
    """Write the four input files with the reference's dataset names and dtypes."""
    data_dir = Path(data_dir)
    data_dir.mkdir(parents=True, exist_ok=True)
    for fname, dsets in sim.items():
        path = data_dir / f"{fname}.h5"
        first = True
        for name, arr in dsets.items():
            a = np.asarray(arr)
            h5io.write(path, name, a, append=not first, gzip=gzip if a.ndim > 0 else 0)
            first = False
    return data_dir


def read_folder(data_dir):
    """Read a sim_data folder (written by the reference's sim_setup or by write_folder) into the dict form."""
    data_dir = Path(data_dir)
    sim = {"sim_consts": {}, "vox_out": {}, "comms_out": {}, "sim_mats": {}}
    need = {
        "sim_consts": ["l", "l2", "Ts", "fcc_flag"],
        "vox_out": ["Nx", "Ny", "Nz", "Nb", "bn_ixyz", "adj_bn", "mat_bn", "saf_bn"],
        "comms_out": ["Nt", "Ns", "Nr", "diff", "in_ixyz", "out_ixyz", "out_reorder", "in_sigs"],
        "sim_mats": ["Nmat", "Mb"],
    }
    opt = {"sim_consts": ["c", "h", "SR", "Tc", "rh"], "vox_out": ["xv", "yv", "zv"], "comms_out": ["out_alpha"]}
    for f, names in need.items():
        p = data_dir / f"{f}.h5"
        if not p.exists():
            raise FileNotFoundError(f"{p} doesn't exist!")  # fdtd_data.h:143 (check_file_exists)
        for n in names:
            sim[f][n] = h5io.read(p, n)
        for n in opt.get(f, []):
            if h5io.exists(p, n):
                sim[f][n] = h5io.read(p, n)
        for n in h5io.list_datasets(p):  # everything else travels along unchanged (the reference edits the files in
            if n not in sim[f]:          # place, rotate_sim_data.py:104-130: e.g. vox_out::h written by VoxScene.save)
                try:
                    sim[f][n] = h5io.read(p, n)
                except TypeError:
                    pass                 # a dataset class this shim cannot carry (strings): left behind
    p = data_dir / "sim_mats.h5"
    for i in range(int(sim["sim_mats"]["Nmat"])):
        sim["sim_mats"][f"mat_{i:02d}_DEF"] = h5io.read(p, f"mat_{i:02d}_DEF")
    return sim
