"""GPU: the energy-conservation diagnostic (pf_engine_run_energy) against golden vectors captured from the
reference PYTHON engine (tests/golden/energy_*.npz, made by make_golden_energy.py through test-only shims).
The Python engine orders its arithmetic differently from the C engines (explicit Laplacian grid, FD update
algebraically rearranged: sim_fdtd.py:816-837), so agreement is to round-off, not bit-exact:
  * receiver outputs: 1e-11 of peak;  * H_tot, E_lost, E_in series: 1e-9 of the series' peak;
  * the balance (H_tot+E_lost-E_in)/2^floor(log2) stays at round-off (<= 5e-13), like the reference's ~1e-15.
"""
from pathlib import Path

import numpy as np
import pytest

from pffdtd_amd import engine, sim_data, synth
from pffdtd_amd.sim_fdtd import rel_diff

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"
import sys
sys.path.insert(0, str(GOLDEN))
from make_golden import digest  # noqa: E402

from energy_cases import ENERGY_CASES  # noqa: E402


@pytest.mark.parametrize("name", list(ENERGY_CASES))
def test_energy_series_match_python_reference(name):
    g = np.load(GOLDEN / f"energy_{name}.npz")
    sim = synth.shoebox(**ENERGY_CASES[name])
    assert digest(sim) == str(g["digest"])
    sd = sim_data.SimData.from_sim(sim, "double")  # no scale_input: the Python engine does not rescale
    eng = engine.HipEngine(sd, energy=True)
    eng.energy_cfg(sd.h, sd.c, sd.Ts, sd.DEF)
    H, El, Ei = np.zeros(sd.Nt), np.zeros(sd.Nt + 1), np.zeros(sd.Nt + 1)
    eng.run_energy(0, sd.Nt, H, El, Ei)
    eng.close()
    peak = np.abs(g["u_out"]).max()
    assert np.abs(sd.u_out - g["u_out"]).max() <= 1e-11 * peak
    for mine, ref, nm in ((H, g["H_tot"], "H_tot"), (El, g["E_lost"], "E_lost"), (Ei, g["E_in"], "E_in")):
        scale = max(np.abs(ref).max(), 1e-300)
        assert np.abs(mine - ref).max() <= 1e-9 * scale, nm
    bal = rel_diff(H + El[:-1], Ei[:-1])
    assert np.abs(bal).max() <= 5e-13, f"energy balance {np.abs(bal).max():.3e}"
    assert np.abs(H).max() > 0 and Ei[-1] != 0


def test_energy_requires_flag():
    sd = sim_data.SimData.from_sim(synth.shoebox(**ENERGY_CASES["cart_lossy"]), "double")
    eng = engine.HipEngine(sd)
    with pytest.raises(engine.PfError):
        eng.energy_cfg(sd.h, sd.c, sd.Ts, sd.DEF)
    eng.close()


def test_baseline_config0_ctk_cart_viz_energy_balance(tmp_path):
    """BASELINE.json configs[0] itself: the CTK church of test_script_CTK_cart_viz.py (fmax 500 Hz, PPW 7.5, 0.1 s: grid
    234 x 154 x 85, fp64, dhann30 input, not differentiated), built from the scene export by sim_setup (device voxelizer),
    all 651 steps with the energy diagnostic on: the normalised balance H_tot + E_lost - E_in must stay at round-off
    (the reference prints the same quantity, sim_fdtd.py:671-678, at ~1e-15 with its summation order)."""
    from pffdtd_amd import scenes
    from pffdtd_amd.sim_setup import sim_setup
    mats = scenes.write_materials(tmp_path / "materials")
    folder = tmp_path / "cfg0"
    sim_setup(**scenes.setup_kwargs("ctk_cart_viz", folder, mats, save_folder_gpu=folder, compress=0))
    sd = sim_data.SimData.from_folder(folder, "double", build_mask=False)
    assert (sd.Nx, sd.Ny, sd.Nz) == (234, 154, 85) and sd.Nt == 651 and sd.fcc_flag == 0
    eng = engine.HipEngine(sd, energy=True)
    eng.energy_cfg(sd.h, sd.c, sd.Ts, sd.DEF)
    H, El, Ei = np.zeros(sd.Nt), np.zeros(sd.Nt + 1), np.zeros(sd.Nt + 1)
    eng.run_energy(0, sd.Nt, H, El, Ei)
    eng.close()
    bal = rel_diff(H + El[:-1], Ei[:-1])
    assert np.isfinite(sd.u_out).all() and np.abs(sd.u_out).max() > 0
    assert Ei[-1] > 0 and El[-1] > 0 and H.max() > 0          # energy went in, some was absorbed by the walls
    assert np.abs(bal).max() <= 1e-12, f"energy balance {np.abs(bal).max():.3e}"
