"""Deterministic un-prepared sim folders (dict form) for the GPU-preparation parity test: axes not in descending order,
boundary nodes / sources / receivers shuffled.  Used by tests/golden/make_golden_prep.py (reference side) and
tests/test_setup_io.py (this package's side)."""
import numpy as np

from pffdtd_amd import synth

# tag: (Nx, Ny, Nz, fcc) -- the rotation must bring them to descending order; FCC grids have even sizes
CASES = {"cart_zxy": (14, 22, 30, False), "cart_yxz": (20, 26, 12, False), "fcc_zyx": (12, 18, 24, True),
         "fcc_xzy": (26, 12, 16, True), "fcc_sorted_dims": (24, 20, 14, True)}


def make(tag):
    Nx, Ny, Nz, fcc = CASES[tag]
    sim = synth.shoebox(Nx, Ny, Nz, Nt=12, fcc=fcc, Nm=2, Mb=[2, 3])
    rng = np.random.default_rng(sum(map(ord, tag)))
    v, c = sim["vox_out"], sim["comms_out"]
    p = rng.permutation(int(v["Nb"]))
    for k in ("bn_ixyz", "adj_bn", "mat_bn", "saf_bn"):
        v[k] = np.ascontiguousarray(v[k][p])
    p = rng.permutation(int(c["Ns"]))
    c["in_ixyz"], c["in_sigs"] = c["in_ixyz"][p], np.ascontiguousarray(c["in_sigs"][p])
    # receivers: shuffle the nodes, and keep the (Nr/8, 8) weight table aligned with them
    p = rng.permutation(int(c["Nr"]))
    c["out_ixyz"] = c["out_ixyz"][p]
    c["out_alpha"] = np.ascontiguousarray(c["out_alpha"].reshape(-1)[p].reshape(c["out_alpha"].shape))
    return sim
