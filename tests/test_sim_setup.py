"""End-to-end `sim_setup` (scene export -> sim folder) against the reference's own sim_setup() output on coarse
versions of its test-script configurations (tests/golden/setup_e2e_*.npz, made by make_golden_setup_e2e.py), then the
folder is GPU-prepared and run on the HIP engine against the CPU oracle."""
from pathlib import Path

import numpy as np
import pytest

from pffdtd_amd import h5io, scenes

GOLD = Path(__file__).resolve().parent / "golden"
DATA = Path(__file__).resolve().parent.parent / "pffdtd_amd" / "data"
CASES = {
    "ctk_cart": ("ctk_cart_viz", dict(duration=0.02, PPW=6.0, fmax=250.0)),
    "ctk_fcc": ("ctk_cart_gpu", dict(source_num=2, duration=0.02, fcc_flag=True, PPW=6.0, fmax=350.0)),
    "mv_fcc": ("mv_fcc_gpu", dict(duration=0.01, PPW=5.0, fmax=500.0)),
}


def test_scene_configs_name_the_reference_test_scripts():
    assert set(scenes.CONFIGS) == {"ctk_cart_viz", "ctk_cart_gpu", "mv_fcc_gpu", "mv_fcc_viz"}
    assert scenes.model_path("CTK").exists() and scenes.model_path("MV").exists()
    z = np.load(DATA / "materials_DEF.npz")
    for cfg in scenes.CONFIGS.values():
        assert set(cfg["mat_files_dict"].values()) <= set(z.files)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", [t for t in CASES if (GOLD / f"setup_e2e_{t}.npz").exists()])
def test_sim_setup_matches_reference_folder(tag, tmp_path):
    from pffdtd_amd.sim_setup import sim_setup
    g = np.load(GOLD / f"setup_e2e_{tag}.npz")
    name, override = CASES[tag]
    mats = scenes.write_materials(tmp_path / "materials")
    folder = tmp_path / "sim"
    sim_setup(**scenes.setup_kwargs(name, folder, mats, compress=0, **override))
    for key in g.files:
        f, ds = key.split("/")
        got = np.asarray(h5io.read(folder / f"{f}.h5", ds))
        want = g[key]
        if key == "vox_out/adj_bn":
            got = np.packbits(got.astype(bool), axis=1, bitorder="little")
        assert got.shape == want.shape, key
        assert np.array_equal(got, want), key  # bit-exact, doubles included


@pytest.mark.gpu
def test_setup_to_engine_end_to_end(tmp_path):
    """scene export -> sim_setup (+GPU prep) -> HIP engine, against the CPU oracle on the same folder."""
    import oracle
    from pffdtd_amd import engine, sim_data
    from pffdtd_amd.sim_setup import sim_setup
    mats = scenes.write_materials(tmp_path / "materials")
    folder, gpu = tmp_path / "sim", tmp_path / "gpu"
    sim_setup(**scenes.setup_kwargs("ctk_cart_gpu", folder, mats, save_folder_gpu=gpu, compress=0, duration=0.03, PPW=6.0, fmax=300.0))
    for prec in ("single", "double"):
        sd = sim_data.SimData.from_folder(gpu, prec)
        sd.scale_input()
        ref = sim_data.SimData.from_folder(gpu, prec)
        ref.scale_input()
        oracle.run_sim(ref)
        engine.run_sim(sd)
        assert np.abs(ref.u_out).max() > 0
        assert np.array_equal(sd.u_out, ref.u_out), prec


@pytest.mark.gpu
def test_musikverein_fcc_setup_to_engine_against_the_oracle(tmp_path):
    """BASELINE configs[2] geometry (python/test_script_MV_fcc_gpu.py:30-38 at a coarse fmax): Musikverein export ->
    sim_setup on the device (13-point FCC, 5 materials x 11 branches) -> fold + sort -> HIP engine, fp32 and fp64,
    bit for bit against the CPU oracle on the same folder, run long enough that the wave reaches every receiver."""
    import oracle
    from pffdtd_amd import engine, sim_data
    from pffdtd_amd.sim_setup import sim_setup
    mats = scenes.write_materials(tmp_path / "materials")
    folder, gpu = tmp_path / "sim", tmp_path / "gpu"
    # The hall's receivers stand as close as 17 cm to a surface (R9; R12 / R13: 21 cm): the grid must be finer than ~6 cm or
    # one of their eight nodes lands on a boundary node, which the reference refuses too (sim_comms.py:233-249).
    for fmax in (1100.0, 1200.0, 1300.0):  # h = 5.7, 5.2, 4.8 cm: 4.0e7 FCC nodes at the first
        try:
            sim_setup(**scenes.setup_kwargs("mv_fcc_gpu", folder, mats, save_folder_gpu=gpu, compress=0, duration=0.09, PPW=5.5, fmax=fmax))
            break
        except AssertionError as exc:
            if "boundary node" not in str(exc):
                raise
    else:
        pytest.fail("no clash-free resolution among the candidates")
    for prec in ("single", "double"):
        sd = sim_data.SimData.from_folder(gpu, prec)
        assert sd.fcc_flag == 2 and sd.Nx >= sd.Nz and sd.Nm == 5 and (np.asarray(sd.Mb) == 11).all()  # (Ny was halved by the fold)
        sd.scale_input()
        ref = sim_data.SimData.from_folder(gpu, prec)
        ref.scale_input()
        oracle.run_sim(ref)
        engine.run_sim(sd)
        heard = np.abs(ref.u_out).max(axis=1) > 0  # 0.09 s = 31 m of travel: all but the receivers at the far end of the hall
        assert heard.mean() >= 0.7, f"{(~heard).sum()} of {heard.size} receiver nodes silent after {sd.Nt} steps"
        assert np.array_equal(sd.u_out, ref.u_out), prec


@pytest.mark.gpu
def test_ctk_church_in_temporally_blocked_pairs_against_the_oracle(tmp_path):
    """BASELINE configs[1] geometry (python/test_script_CTK_cart_gpu.py:33-38 at a coarse fmax) with pairs FORCED
    (air_variant 40): the clean tiles of the church step two at a time, the tiles holding walls, pews and the source one
    at a time -- receivers bit for bit the CPU oracle's."""
    import oracle
    from pffdtd_amd import engine, sim_data
    from pffdtd_amd.sim_setup import sim_setup
    mats = scenes.write_materials(tmp_path / "materials")
    folder, gpu = tmp_path / "sim", tmp_path / "gpu"
    sim_setup(**scenes.setup_kwargs("ctk_cart_gpu", folder, mats, save_folder_gpu=gpu, compress=0, duration=0.035, PPW=6.0, fmax=640.0))
    for prec in ("single", "double"):
        ref = sim_data.SimData.from_folder(gpu, prec)
        ref.scale_input()
        oracle.run_sim(ref)
        assert np.abs(ref.u_out).max() > 0
        sd = sim_data.SimData.from_folder(gpu, prec)
        sd.scale_input()
        eng = engine.HipEngine(sd, air_variant=40, timing=True)
        eng.run(0, sd.Nt)
        tm = eng.timing()
        eng.close()
        assert tm["tb2_launches"] > 0 and tm["tb2_cells"] > 0, (tm, sd.Nx, sd.Ny, sd.Nz)  # (few tiles are clean at this resolution)
        assert np.array_equal(sd.u_out, ref.u_out), prec
