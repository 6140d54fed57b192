#!/usr/bin/env python3
"""From a scene export to room impulse responses, entirely through this package on one MI355X:

    wall-impedance tables (shipped DEF fits)  ->  sim_setup (device voxelizer, GPU prep)  ->  HIP engine  ->  process_outputs

The same chain as the reference's  test_script_CTK_cart_gpu.py -> fdtd_main_gpu_single.x ->
fdtd.process_outputs , here on the CTK church export at a reduced bandwidth so it finishes in seconds
(--fmax 1400 --duration 3.0 is the reference's full configuration: 894x579x309 grid, 76 460 steps, about a minute).

    python examples/ctk_rir.py --out /tmp/ctk --fmax 500 --duration 0.25 [--air_abs stokes] [--wav]
"""
import argparse
import sys
import time
from pathlib import Path


ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pffdtd_amd import engine, scenes, sim_data  # noqa: E402
from pffdtd_amd.process_outputs import ProcessOutputs  # noqa: E402
from pffdtd_amd.sim_setup import sim_setup  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--fmax", type=float, default=500.0)
    ap.add_argument("--ppw", type=float, default=10.5)
    ap.add_argument("--duration", type=float, default=0.25)
    ap.add_argument("--precision", default="single", choices=["single", "double"])
    ap.add_argument("--air_abs", default="none", choices=["none", "stokes", "modal", "ola"])
    ap.add_argument("--wav", action="store_true")
    a = ap.parse_args(argv)
    out = Path(a.out)
    t0 = time.perf_counter()
    mats = scenes.write_materials(out / "materials")  # 1. wall impedances: the reference's fitted DEF tables (data files)
    t1 = time.perf_counter()
    folder = out / "sim"  # 2. grid, sources/receivers, voxelization, GPU prep
    sim_setup(**scenes.setup_kwargs("ctk_cart_gpu", folder, mats, save_folder_gpu=folder, compress=0, fmax=a.fmax, PPW=a.ppw,
                                    duration=a.duration, diff_source=True))
    t2 = time.perf_counter()
    sd = sim_data.SimData.from_folder(folder, a.precision, build_mask=False)  # 3. time stepping (fdtd_main.c:44-53)
    sd.scale_input()
    engine.run_sim(sd)
    sd.rescale_output()
    sd.write_outputs(folder)
    t3 = time.perf_counter()
    po = ProcessOutputs(folder)  # 4. receivers -> RIRs
    po.initial_process(fcut=10.0, N_order=4)
    po.apply_lowpass(fcut=a.fmax, N_order=8, symmetric=True)
    po.resample(48e3)
    if a.air_abs != "none":
        {"stokes": po.apply_stokes_filter, "modal": po.apply_modal_filter, "ola": po.apply_ola_filter}[a.air_abs]()
    po.save_h5()
    if a.wav:
        po.save_wav()
    t4 = time.perf_counter()
    print(f"grid {sd.Nx}x{sd.Ny}x{sd.Nz}, {sd.Nt} steps, {sd.Nr // 8} receivers: materials {t1-t0:.1f} s, setup {t2-t1:.1f} s, "
          f"engine {t3-t2:.1f} s ({sd.Npts * sd.Nt / (t3 - t2) / 1e9:.0f} Gvox/s incl. load), post {t4-t3:.1f} s")
    return po


if __name__ == "__main__":
    main()
