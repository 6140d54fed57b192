"""GPU: the two command-line hosts end to end on a sim_data folder (file contract in, sim_outs.h5 out)."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import cases
import oracle
from pffdtd_amd import h5io, sim_data, synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _expected(folder, prec, scale):
    sd = sim_data.SimData.from_folder(folder, prec)
    if scale:
        sd.scale_input()
    oracle.run_sim(sd)
    if scale:
        sd.rescale_output()
    return sd.u_out[sd.out_reorder, :]


def test_fdtd_main_cli(tmp_path):
    sim = synth.sort_sim(cases.make_sim("cart_outside"))
    synth.write_folder(sim, tmp_path, gzip=3)
    r = subprocess.run([sys.executable, "-m", "pffdtd_amd.fdtd_main", "--precision", "single"], cwd=tmp_path,
                       env={**__import__("os").environ, "PYTHONPATH": str(ROOT)}, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    for line in ("Air update:", "Boundary loop:", "Combined (total):", "RAW OUTPUTS", "wrote output dataset"):
        assert line in r.stdout
    assert np.array_equal(h5io.read(tmp_path / "sim_outs.h5", "u_out"), _expected(tmp_path, "single", True))


def test_sim_fdtd_cli_with_energy(tmp_path):
    sim = synth.shoebox(Nx=16, Ny=14, Nz=12, Nt=40, Nm=2, Mb=[2, 3], diff=False, sig="dhann30")
    synth.write_folder(sim, tmp_path)
    r = subprocess.run([sys.executable, "-m", "pffdtd_amd.sim_fdtd", "--data_dir", str(tmp_path), "--energy", "--nsteps", "7"],
                       env={**__import__("os").environ, "PYTHONPATH": str(ROOT)}, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "normalised energy balance:" in r.stdout and "GRID OUTPUTS" in r.stdout
    bal = [float(l.split(":")[-1]) for l in r.stdout.splitlines() if "normalised energy balance" in l]
    assert len(bal) == 5 and max(abs(b) for b in bal) < 1e-12
    # energy mode runs the unfused kernels in the reference order: receivers are still the C-engine bits
    assert np.array_equal(h5io.read(tmp_path / "sim_outs.h5", "u_out"), _expected(tmp_path, "double", False))


@pytest.mark.parametrize("gpus,name,prec", [(2, "cart_outside", "single"), (3, "fcc2_outside", "double")])
def test_fdtd_main_cli_under_an_external_launcher(tmp_path, gpus, name, prec):
    """One process per GPU under `torch.distributed.run` (WORLD_SIZE set): Z-slabs cut by slab.py, plane exchange through
    torch.distributed, rank 0 writes sim_outs.h5.  Here all ranks share GPU 0 and the planes go through gloo
    (PFFDTD_BACKEND=gloo); the folder's lists are left unsorted on purpose (the reference's multi-GPU engine would refuse
    them, gpu_engine.h:688)."""
    import os
    sim = cases.make_sim(name)
    synth.write_folder(sim, tmp_path)
    env = {**os.environ, "PYTHONPATH": str(ROOT), "PFFDTD_BACKEND": "gloo"}
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
                        "--master-port", "29653", "-m", "pffdtd_amd.fdtd_main", "--precision", prec, "--data_dir", str(tmp_path)],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    for line in (f"--{gpus} GPUs", "Air update:", "Combined (total):", "RAW OUTPUTS", "wrote output dataset"):
        assert line in r.stdout, r.stdout[-2000:]
    assert np.array_equal(h5io.read(tmp_path / "sim_outs.h5", "u_out"), _expected(tmp_path, prec, True))


def test_fdtd_main_cli_gpus_needs_that_many_devices(tmp_path):
    """`--gpus N` drives devices 0 .. N-1 from this one process through the C chain (round 4): on a box with fewer devices it
    says so and points at --devices (virtual slabs) instead of silently doing something else."""
    import os
    from pffdtd_amd import engine
    sim = cases.make_sim("cart_outside")
    synth.write_folder(sim, tmp_path)
    n = engine.device_count() + 1
    r = subprocess.run([sys.executable, "-m", "pffdtd_amd.fdtd_main", "--precision", "single", "--gpus", str(n)], cwd=tmp_path,
                       env={**os.environ, "PYTHONPATH": str(ROOT)}, capture_output=True, text=True)
    assert r.returncode != 0 and "--devices" in (r.stdout + r.stderr)


def test_fdtd_main_cli_progress_lines(tmp_path):
    """`--progress K`: the reference's progress fields (fdtd_common.h:106-190) every K steps, one line per report"""
    import os
    sim = cases.make_sim("cart_outside")
    synth.write_folder(sim, tmp_path)
    r = subprocess.run([sys.executable, "-m", "pffdtd_amd.fdtd_main", "--precision", "double", "--devices", "0,0", "--progress", "20"], cwd=tmp_path,
                       env={**os.environ, "PYTHONPATH": str(ROOT)}, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = [ln for ln in r.stdout.splitlines() if ln.startswith("Running [")]
    assert len(rep) == 4 and rep[-1].startswith("Running [100.0%]")  # 70 steps: 20, 40, 60, 70
    for f in ("T: ", "I: ", "TPW: ", "IPW: ", "TA: ", "IA: ", "TB: ", "IB: "):
        assert f in rep[0], rep[0]
    assert np.array_equal(h5io.read(tmp_path / "sim_outs.h5", "u_out"), _expected(tmp_path, "double", True))


def test_fdtd_main_cli_in_process_device_chain(tmp_path):
    """`--devices 0,0,0`: one process, the C library's multi-device run_sim (three slabs on GPU 0), unsorted lists"""
    import os
    sim = cases.make_sim("cart_outside")
    synth.write_folder(sim, tmp_path)
    r = subprocess.run([sys.executable, "-m", "pffdtd_amd.fdtd_main", "--precision", "double", "--devices", "0,0,0"], cwd=tmp_path,
                       env={**os.environ, "PYTHONPATH": str(ROOT)}, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "3 slabs" in r.stdout and "wrote output dataset" in r.stdout and "Air update:" in r.stdout and "Combined (total):" in r.stdout
    assert np.array_equal(h5io.read(tmp_path / "sim_outs.h5", "u_out"), _expected(tmp_path, "double", True))
