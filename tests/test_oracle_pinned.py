"""The CPU oracle (oracle/pf_oracle.c) is pinned bit-for-bit to the reference:
   (a) against the golden receiver outputs captured from the compiled reference binaries (tests/golden/*.npz,
       made by tests/golden/make_golden.py) -- runs everywhere;
   (b) against the compiled reference itself (oracle/_ref) where it exists (the build container).
Both go through the whole host flow of c_cuda/fdtd_main.c:44-53: load -> scale_input -> run_sim -> rescale_output
-> write order, so the Python loader (pffdtd_amd/sim_data.py) is pinned too.
"""
import hashlib
import tempfile
from pathlib import Path

import numpy as np
import pytest

import cases
import oracle
from pffdtd_amd import sim_data, synth

GOLDEN = Path(__file__).resolve().parent / "golden"
FIXTURES = sorted(p for p in GOLDEN.glob("*.npz") if p.stem.endswith(("_single", "_double")) and not p.name.startswith("energy_"))


def _digest(sim):
    h = hashlib.sha256()
    for f in sorted(sim):
        for k in sorted(sim[f]):
            a = np.ascontiguousarray(sim[f][k])
            h.update(f"{f}/{k}/{a.dtype}/{a.shape}".encode())
            h.update(a.tobytes())
    return h.hexdigest()


def _host_flow(sim, prec):
    sd = sim_data.SimData.from_sim(sim, prec)
    sd.scale_input()
    oracle.run_sim(sd)
    sd.rescale_output()
    return sd.u_out[sd.out_reorder, :]


def test_fixtures_present():
    assert len(FIXTURES) >= 16


@pytest.mark.parametrize("fx", FIXTURES, ids=lambda p: p.stem)
def test_oracle_matches_golden(fx):
    name, prec = fx.stem.rsplit("_", 1)
    g = np.load(fx)
    sim = cases.make_sim(name)
    assert _digest(sim) == str(g["digest"]), "scene generator drifted: regenerate tests/golden with make_golden.py"
    u = _host_flow(sim, prec)
    assert u.shape == g["u_out"].shape
    assert np.array_equal(u, g["u_out"]), f"max|d|={np.abs(u - g['u_out']).max()}"


@pytest.mark.skipif(oracle.ref_binary("double") is None, reason="compiled reference (oracle/_ref) not available")
@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("name", ["cart_rigid", "cart_oddz", "fcc1_outside", "fcc2_lossy"])
def test_oracle_matches_compiled_reference(name, prec, tmp_path):
    sim = cases.make_sim(name)
    synth.write_folder(sim, tmp_path, gzip=3)  # the reference reads gzip'ed datasets too
    ref, log = oracle.run_reference(tmp_path, prec, threads=4)
    u = _host_flow(synth.read_folder(tmp_path), prec)
    assert np.array_equal(u, ref)
    assert "Combined (total)" in log


@pytest.mark.skipif(oracle.ref_binary("double") is None, reason="compiled reference (oracle/_ref) not available")
def test_sim_outs_file_roundtrip(tmp_path):
    """write_outputs() writes what the reference writes (sim_outs.h5::u_out f64[Nr,Nt], reordered rows)."""
    from pffdtd_amd import h5io
    sim = cases.make_sim("cart_lossy")
    synth.sort_sim(sim)  # non-trivial out_reorder
    synth.write_folder(sim, tmp_path)
    ref, _ = oracle.run_reference(tmp_path, "double", threads=2)
    sd = sim_data.SimData.from_folder(tmp_path, "double")
    sd.scale_input()
    oracle.run_sim(sd)
    sd.rescale_output()
    sd.write_outputs(tmp_path)
    assert np.array_equal(h5io.read(tmp_path / "sim_outs.h5", "u_out"), ref)


def test_fold_is_equivalent_in_double():
    """Folding the FCC subgrid (fcc_flag 1 -> 2) must not change the receivers beyond round-off."""
    sim1 = synth.shoebox(24, 28, 20, Nt=80, fcc=True, Nm=2, Mb=[2, 3])
    sim2 = synth.fold_fcc(synth.shoebox(24, 28, 20, Nt=80, fcc=True, Nm=2, Mb=[2, 3]))
    u1, u2 = _host_flow(sim1, "double"), _host_flow(sim2, "double")
    assert np.abs(u1 - u2).max() <= 1e-12 * np.abs(u1).max()


@pytest.mark.parametrize("fx", sorted(GOLDEN.glob("energy_*.npz")), ids=lambda p: p.stem)
def test_oracle_agrees_with_python_reference_engine(fx):
    """C-ordered oracle (fp64, unscaled input) vs the reference PYTHON engine's receivers (captured through
    test-only shims by tests/golden/make_golden_energy.py): same scheme, different association -> round-off only
    (1e-14 absolute after 60 steps).  Tolerance 1e-11 of peak."""
    import sys
    sys.path.insert(0, str(GOLDEN))
    from energy_cases import ENERGY_CASES
    g = np.load(fx)
    name = fx.stem[len("energy_"):]
    sim = synth.shoebox(**ENERGY_CASES[name])
    assert _digest(sim) == str(g["digest"])
    sd = sim_data.SimData.from_sim(sim, "double")
    oracle.run_sim(sd)
    assert np.abs(sd.u_out - g["u_out"]).max() <= 1e-11 * np.abs(g["u_out"]).max()


@pytest.mark.parametrize("fcc", [False, True])
def test_rotation_is_equivalent(fcc):
    """rotate_sim (rotate_sim_data.py:30-130): the physics does not care which axis is the slab axis."""
    kw = dict(Nx=18, Ny=26, Nz=22, Nt=60, fcc=fcc, Nm=2, Mb=[2, 3], wall=6, src=[2, 20, 2], rcv=[[12, 2, 17], [2, 2, 2]])
    a = synth.shoebox(**kw)
    b = synth.rotate_sim(synth.shoebox(**kw))
    assert (int(b["vox_out"]["Nx"]), int(b["vox_out"]["Ny"]), int(b["vox_out"]["Nz"])) == (26, 22, 18)
    ua, ub = _host_flow(a, "double"), _host_flow(b, "double")
    assert np.abs(ua).max() > 0
    assert np.abs(ua - ub).max() <= 1e-12 * np.abs(ua).max()
