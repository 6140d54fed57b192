"""Setup-side writers (pffdtd_amd/setup_io.py) against golden values captured from the reference's own
SimConsts / SimComms classes (tests/golden/make_golden_setup.py), plus folder round trips of prep_folder."""
from pathlib import Path

import numpy as np
import pytest

import cases
import oracle
from pffdtd_amd import h5io, setup_io, sim_data, synth

G = np.load(Path(__file__).resolve().parent / "golden" / "setup_reference.npz")


@pytest.mark.parametrize("tag,fcc", [("cart", False), ("fcc", True)])
def test_consts_and_comms_match_reference(tag, fcc):
    sc = setup_io.SimConsts(Tc=20, rh=50, fmax=700.0, PPW=8.0, fcc=fcc)
    for k in ("h", "c", "Ts", "SR", "l", "l2"):
        assert getattr(sc, k) == G[f"{tag}_{k}"], k
    N = (20, 18, 16)
    xv, yv, zv = (np.arange(n) * sc.h + o for n, o in zip(N, (-0.3, 0.1, 0.05)))
    cm = setup_io.SimComms(h=sc.h, Ts=sc.Ts, l2=sc.l2, fcc_flag=int(fcc), xv=xv, yv=yv, zv=zv)
    cm.prepare_source_pts(G[f"{tag}_S"])
    cm.prepare_receiver_pts(G[f"{tag}_R"])
    cm.prepare_source_signals(60 * sc.Ts, sig_type="dhann30")
    assert np.array_equal(cm.in_ixyz, G[f"{tag}_in_ixyz"]) and np.array_equal(cm.out_ixyz, G[f"{tag}_out_ixyz"])
    assert np.array_equal(cm.in_alpha, G[f"{tag}_in_alpha"]) and np.array_equal(cm.out_alpha, G[f"{tag}_out_alpha"])
    assert np.array_equal(cm.in_sigs, G[f"{tag}_in_sigs"])
    cm.diff_source()
    ref = G[f"{tag}_in_sigs_diff"]  # scipy.signal.lfilter in the reference: same recurrence, allow its rounding order
    assert np.abs(cm.in_sigs - ref).max() <= 1e-12 * np.abs(ref).max()


def test_setup_folder_runs_through_the_engine_flow(tmp_path):
    """consts + grid + comms + mats written by setup_io, boundary nodes by the synthetic generator: a complete
    sim_data folder that the loader accepts and the oracle runs."""
    sc = setup_io.SimConsts(Tc=20, rh=50, h=0.05, fcc=False)
    sc.save(tmp_path)
    cg = setup_io.CartGrid(h=sc.h, offset=3.5, bmin=[0, 0, 0], bmax=[0.9, 0.8, 0.7])
    cg.save(tmp_path)
    cm = setup_io.SimComms(save_folder=tmp_path)
    cm.prepare_source_pts(np.array([0.45, 0.4, 0.35]))
    cm.prepare_receiver_pts(np.array([[0.2, 0.2, 0.2], [0.7, 0.6, 0.5]]))
    cm.prepare_source_signals(40 * sc.Ts, sig_type="impulse")
    cm.diff_source()
    cm.save()
    box = synth.shoebox(cg.Nx, cg.Ny, cg.Nz, Nt=40, Nm=2, Mb=[3, 2])
    for name in ("mat_00", "mat_01"):
        h5io.write(tmp_path / f"{name}.h5", "DEF", box["sim_mats"][f"{name}_DEF"], append=False)
    setup_io.SimMats(tmp_path).package({"a_wall": "mat_00.h5", "b_floor": "mat_01.h5"}, ["a_wall", "_RIGID", "b_floor"], tmp_path)
    first = True
    for k, v in box["vox_out"].items():
        h5io.write(tmp_path / "vox_out.h5", k, v, append=not first)
        first = False
    cm.check_for_clashes(box["vox_out"]["bn_ixyz"])
    sd = sim_data.SimData.from_folder(tmp_path, "single")
    sd.scale_input()
    oracle.run_sim(sd)
    assert sd.Nr == 16 and sd.Ns == 8 and np.isfinite(sd.u_out).all() and np.abs(sd.u_out).max() > 0


def test_prep_folder_matches_in_memory_transforms(tmp_path):
    sim = cases.make_sim("fcc1_outside")
    synth.write_folder(sim, tmp_path / "a")
    setup_io.prep_folder(tmp_path / "a", out_dir=tmp_path / "b")
    got = synth.read_folder(tmp_path / "b")
    want = synth.sort_sim(synth.fold_fcc(synth.rotate_sim(cases.make_sim("fcc1_outside"))))
    assert int(got["sim_consts"]["fcc_flag"]) == 2
    for f in ("vox_out", "comms_out"):
        for k in ("bn_ixyz", "adj_bn", "in_ixyz", "out_ixyz", "out_reorder", "in_sigs"):
            if k in want[f]:
                assert np.array_equal(np.asarray(got[f][k]), np.asarray(want[f][k])), (f, k)


@pytest.mark.parametrize("diff", [True, False])
def test_process_outputs_matches_reference(diff, tmp_path):
    """Recombination, integrator + low-cut and symmetric low-pass vs the reference's ProcessOutputs methods
    (golden: tests/golden/post_reference.npz, produced by running the reference class behind a fake h5py)."""
    from pffdtd_amd import process_outputs
    g = np.load(Path(__file__).resolve().parent / "golden" / "post_reference.npz")
    Ts = float(g["post_Ts"])
    h5io.write(tmp_path / "sim_consts.h5", "Ts", np.float64(Ts), append=False)
    c = tmp_path / "comms_out.h5"
    h5io.write(c, "out_alpha", g["post_alpha"], append=False)
    h5io.write(c, "Nr", np.int64(16)); h5io.write(c, "Nt", np.int64(400)); h5io.write(c, "diff", np.int8(diff))
    h5io.write(tmp_path / "sim_outs.h5", "u_out", g["post_u_out"], append=False)
    po = process_outputs.ProcessOutputs(tmp_path)
    po.initial_process(fcut=10.0, N_order=4)
    assert np.array_equal(po.r_out, g["post_r_out"])
    assert np.array_equal(h5io.read(tmp_path / "sim_outs.h5", "r_out"), g["post_r_out"])       # appended next to u_out
    assert np.array_equal(h5io.read(tmp_path / "sim_outs.h5", "u_out"), g["post_u_out"])       # and u_out kept
    assert np.allclose(po.r_out_f, g[f"post_r_out_f_diff{int(diff)}"], rtol=1e-12, atol=0)
    po.apply_lowpass(fcut=4000.0, N_order=8, symmetric=True)
    assert np.allclose(po.r_out_f, g[f"post_lowpass_diff{int(diff)}"], rtol=1e-12, atol=1e-300)
    po.resample(48e3)
    assert abs(po.Fs_f - 48e3) < 1e-6 and po.r_out_f.shape == (2, 768)
    po.save_h5()
    assert h5io.read(tmp_path / "sim_outs_processed.h5", "r_out_f").shape == (2, 768)


@pytest.mark.parametrize("tag", ["cart_zxy", "cart_yxz", "fcc_zyx", "fcc_xzy", "fcc_sorted_dims"])
def test_gpu_prep_matches_reference_rotate_sim_data(tag):
    """rotate -> fold -> sort against the reference's own rotate_sim_data.py functions run on the same un-prepared folder
    (tests/golden/prep_reference_*.npz, made by make_golden_prep.py): every dataset they touch, bit for bit."""
    import prep_cases
    g = np.load(Path(__file__).resolve().parent / "golden" / f"prep_reference_{tag}.npz")
    sim = prep_cases.make(tag)
    synth.rotate_sim(sim)
    if int(sim["sim_consts"]["fcc_flag"]) == 1:
        synth.fold_fcc(sim)
    synth.sort_sim(sim)
    for key in g.files:
        f, k = key.split("/")
        got = np.asarray(sim[f][k])
        if k == "adj_bn":
            got = np.packbits(got.astype(bool), axis=1, bitorder="little")
        assert got.shape == g[key].shape and np.array_equal(got, g[key]), key
