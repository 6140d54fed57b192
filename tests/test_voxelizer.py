"""Voxelizer: room geometry, Cartesian grid, and the cut-leg computation.

CPU tests pin the host mirrors (RoomGeo, CartGrid) and the numpy oracle (oracle/vox_oracle.py) bit-exactly against
outputs of the reference voxelizer itself on the CTK church model (tests/golden/vox_*.npz, made by
tests/golden/make_golden_vox.py).  GPU tests run the HIP voxelizer through the C ABI against the same goldens
(bit-exact: boundary node set, adjacency bits, materials, surface-area factors), against the oracle on seeded random
triangle soups, and at BASELINE cfg2 resolution (894x579x309) against SHA-256 digests of the reference's output.
"""
import hashlib
from pathlib import Path

import numpy as np
import pytest

import vox_oracle as vo  # oracle/ is on sys.path (tests/conftest.py)
from pffdtd_amd import h5io, setup_io
from pffdtd_amd.room_geo import RoomGeo, tris_precompute

GOLD = Path(__file__).resolve().parent / "golden"
DATA = Path(__file__).resolve().parent.parent / "pffdtd_amd" / "data"
MODELS = {"ctk": DATA / "models" / "CTK_Church_model_export.json", "mv": DATA / "models" / "MV_model_export.json.gz"}
SMALL = ["ctk_cart_h40", "ctk_fcc_h40", "ctk_cart_h25_rot", "ctk_fcc_h30_rot"]
MV = ["mv_fcc_h20", "mv_cart_h25"]  # Musikverein export: 32k triangles (rows of chairs)
# hand-made export (tests/golden/make_open_scene.py): unmarked _RIGID triangles, open top with custom bounds, two-sided
# panel, tilted one-sided reflector; goldens from the reference voxelizer like the others
OPEN = ["open_cart_h10", "open_fcc_h12"]
OPEN_BOUNDS = (np.array([-0.4, -0.4, -0.3]), np.array([4.5, 3.7, 3.4]))
MODELS["open"] = DATA / "models" / "open_scene.json"


def scene(tag):
    g = np.load(GOLD / f"vox_{tag}.npz")
    kw = dict(bmin=OPEN_BOUNDS[0].copy(), bmax=OPEN_BOUNDS[1].copy()) if tag.startswith("open") else {}
    rg = RoomGeo(str(MODELS[tag.split("_")[0]]), az_el=tuple(g["az_el"]), **kw)
    cg = setup_io.CartGrid(h=float(g["h"]), offset=3.5, bmin=rg.bmin, bmax=rg.bmax, fcc=bool(g["fcc"]))
    return g, rg, cg


def bits_of(adj):
    NN = adj.shape[1]
    return (adj.astype(np.uint16) << np.arange(NN, dtype=np.uint16)).sum(axis=1).astype(np.uint16)


@pytest.mark.parametrize("tag", SMALL + ["ctk_cart_h0915"] + MV + OPEN)
def test_room_geo_and_grid_match_reference(tag):
    g, rg, cg = scene(tag)
    assert np.array_equal(rg.bmin, g["bmin"]) and np.array_equal(rg.bmax, g["bmax"])
    assert rg.tris.shape[0] == int(g["ntris"])
    assert rg.vol == float(g["vol"]) and rg.area == float(g["area"])
    assert np.array_equal(rg.mat_area, g["mat_area"])
    assert list(rg.mat_str) == [str(s) for s in g["mat_str"]]
    assert [cg.Nx, cg.Ny, cg.Nz] == g["Nxyz"].tolist()
    assert np.array_equal(np.array([cg.xv[0], cg.yv[0], cg.zv[0]]), g["xyzmin"])


@pytest.mark.parametrize("tag", SMALL + ["ctk_cart_h0915", "mv_cart_h25"] + OPEN)
def test_oracle_pinned_to_reference_voxelizer(tag):
    g, rg, cg = scene(tag)
    fcc, h = bool(g["fcc"]), float(g["h"])
    bn, adj, tidx, _ = vo.calc_adj(cg.xv, cg.yv, cg.zv, h, fcc, rg.tris_pre)
    mat, saf = vo.materials_and_saf(bn, adj, tidx, cg.xv, cg.yv, cg.zv, h, fcc, rg.tris_pre, rg.mat_ind, rg.mat_side)
    assert np.array_equal(bn, g["bn_ixyz"])
    assert np.array_equal(bits_of(adj), g["adj_bits"])
    assert np.array_equal(mat, g["mat_bn"])
    assert np.array_equal(saf, g["saf_bn"])  # bit-exact doubles


def soup(seed, ntri, h, n=(26, 22, 20), big=0.3):
    """Random triangle soup inside a grid (a few large triangles, many small ones, some axis-aligned on grid planes)."""
    rng = np.random.default_rng(seed)
    xv, yv, zv = (np.arange(m) * h + o for m, o in zip(n, (-0.31, 0.07, 1.3)))
    L = np.array([xv[-1] - xv[0], yv[-1] - yv[0], zv[-1] - zv[0]])
    o = np.array([xv[0], yv[0], zv[0]])
    c = o + L * (0.15 + 0.7 * rng.random((ntri, 1, 3)))
    size = np.where(rng.random((ntri, 1, 1)) < big, 0.5, 0.08) * L.min()
    pts = c + size * (rng.random((ntri, 3, 3)) - 0.5)
    k = ntri // 5  # exactly on grid planes / through grid points: exercises the on-surface ("near boundary") branch
    pts[:k, :, 0] = xv[rng.integers(3, n[0] - 3, size=(k, 1))]
    pts[k:2 * k, :, 2] = zv[rng.integers(3, n[2] - 3, size=(k, 1))]
    pts = pts.reshape(-1, 3)
    tris = np.arange(ntri * 3).reshape(-1, 3)
    pre = tris_precompute(pts, tris)
    keep = pre["area"] > 1e-6
    return xv, yv, zv, {kk: v[keep] for kk, v in pre.items()}


def test_oracle_partition_independence():
    """The oracle's answer may not depend on how a triangle's padded box is cut: shuffling triangle order must only
    change which of several equidistant triangles is reported (never the cut legs)."""
    xv, yv, zv, pre = soup(3, 40, 0.1)
    bn, adj, _, nd = vo.calc_adj(xv, yv, zv, 0.1, False, pre)
    perm = np.random.default_rng(0).permutation(pre["cent"].shape[0])
    bn2, adj2, _, nd2 = vo.calc_adj(xv, yv, zv, 0.1, False, {k: v[perm] for k, v in pre.items()})
    assert np.array_equal(bn, bn2) and np.array_equal(adj, adj2) and np.array_equal(nd, nd2)


# ------------------------------------------------------------------ GPU ------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("tag", SMALL + ["ctk_cart_h0915"] + MV + OPEN)
def test_hip_voxelizer_matches_reference(tag):
    from pffdtd_amd.voxelizer import VoxScene
    g, rg, cg = scene(tag)
    vs = VoxScene(rg, cg, fcc=bool(g["fcc"]))
    vs.calc_adj()
    assert np.array_equal(vs.bn_ixyz, g["bn_ixyz"])
    assert np.array_equal(bits_of(vs.adj_bn), g["adj_bits"])
    assert np.array_equal(vs.mat_bn, g["mat_bn"])
    assert np.array_equal(vs.saf_bn, g["saf_bn"])
    if tag.startswith("open"):
        assert (vs.mat_bn == -1).sum() > 500 and rg.mat_str[-1] == "_RIGID"  # the rigid floor
    assert vs.check_adj_full() <= (16 if tag.startswith("mv") else 0)  # the reference's own output; see sim_setup.py


@pytest.mark.gpu
@pytest.mark.parametrize("seed,fcc,ntri", [(1, False, 60), (2, True, 60), (3, False, 400), (4, True, 300), (5, False, 3)])
def test_hip_voxelizer_matches_oracle_on_triangle_soups(seed, fcc, ntri):
    from pffdtd_amd.voxelizer import cut_legs
    h = 0.1
    n = (26, 22, 20) if not fcc else (26, 22, 40)
    xv, yv, zv, pre = soup(seed, ntri, h, n=n)
    bn, adj, tidx, nd = vo.calc_adj(xv, yv, zv, h, fcc, pre)
    bn2, adj2, tidx2, nd2, st = cut_legs(xv, yv, zv, h, fcc, pre)
    assert bn.size > 5
    assert np.array_equal(bn, bn2) and np.array_equal(adj, adj2)
    assert np.array_equal(tidx, tidx2) and np.array_equal(nd, nd2)
    assert st["npairs"] > 0


@pytest.mark.gpu
def test_hip_voxelizer_no_triangles_near_grid():
    from pffdtd_amd.voxelizer import cut_legs
    xv, yv, zv, pre = soup(7, 8, 0.1)
    far = {k: (v + 100.0 if k in ("v", "cent", "bmin", "bmax") else v) for k, v in pre.items()}
    bn, adj, tidx, nd, st = cut_legs(xv, yv, zv, 0.1, False, far)
    assert bn.size == 0 and adj.shape == (0, 6) and st["npairs"] == 0


@pytest.mark.gpu
def test_hip_voxelizer_cfg2_resolution_digests(tmp_path):
    """BASELINE cfg2 grid (CTK church, fmax 1400 Hz, PPW 10.5 -> 894x579x309 = 1.6e8 points): the reference takes
    ~100 s for this on one core; compare SHA-256 of the sorted arrays and write/read vox_out.h5."""
    from pffdtd_amd.voxelizer import VoxScene
    g, rg, cg = scene("ctk_cart_cfg2_digest")
    assert [cg.Nx, cg.Ny, cg.Nz] == g["Nxyz"].tolist() == [894, 579, 309]
    vs = VoxScene(rg, cg, fcc=False)
    vs.calc_adj()
    assert vs.bn_ixyz.size == int(g["Nb"])
    for name, arr in (("bn_ixyz", vs.bn_ixyz), ("adj_bits", bits_of(vs.adj_bn)), ("mat_bn", vs.mat_bn), ("saf_bn", vs.saf_bn)):
        assert hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest() == str(g[name + "_sha256"]), name
    vs.save(tmp_path, compress=1)
    f = tmp_path / "vox_out.h5"
    assert np.array_equal(h5io.read(f, "bn_ixyz"), vs.bn_ixyz)
    assert np.array_equal(h5io.read(f, "adj_bn").astype(bool), vs.adj_bn)
    assert int(h5io.read(f, "Nb")) == vs.bn_ixyz.size and int(h5io.read(f, "Nx")) == 894


@pytest.mark.gpu
def test_hip_voxelizer_mv_viz_resolution_digests():
    """Musikverein at the spacing of test_script_MV_fcc_viz.py (836x328x254 FCC, 1.57 M boundary nodes): SHA-256 of the
    reference's arrays (436 s on one core there)."""
    from pffdtd_amd.voxelizer import VoxScene
    g, rg, cg = scene("mv_fcc_viz_digest")
    assert [cg.Nx, cg.Ny, cg.Nz] == g["Nxyz"].tolist()
    vs = VoxScene(rg, cg, fcc=True)
    vs.calc_adj()
    assert vs.bn_ixyz.size == int(g["Nb"])
    for name, arr in (("bn_ixyz", vs.bn_ixyz), ("adj_bits", bits_of(vs.adj_bn)), ("mat_bn", vs.mat_bn), ("saf_bn", vs.saf_bn)):
        assert hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest() == str(g[name + "_sha256"]), name
