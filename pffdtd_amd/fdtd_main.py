"""Drop-in for the reference binaries `fdtd_main_{cpu,gpu}_{single,double}.x` (c_cuda/fdtd_main.c:35-59).

    cd <sim_data folder> && python -m pffdtd_amd.fdtd_main --precision single [--gpus N]

Same flow: load_sim_data -> scale_input -> run_sim -> rescale_output -> write_outputs -> print_last_samples,
reading the four input .h5 files from the current directory and writing sim_outs.h5 there.
"""
import argparse
import time
from pathlib import Path

from . import engine, sim_data


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--precision", default="single", choices=["single", "double"])
    p.add_argument("--data_dir", default=".", help="folder with the input .h5 files (the reference uses the CWD)")
    p.add_argument("--gpu", type=int, default=0)
    a = p.parse_args()
    print(f"--Date and time: {time.ctime()}")
    sd = sim_data.SimData.from_folder(Path(a.data_dir), a.precision)
    sd.scale_input()
    eng = engine.HipEngine(sd, device=a.gpu, timing=True)
    t0 = time.perf_counter()
    eng.run(0, sd.Nt)
    eng.sync()
    el = time.perf_counter() - t0
    tm = eng.timing()
    eng.close()
    t_air = tm["air_ms_total"] * 1e-3
    t_rest = max(tm["step_ms_total"] * 1e-3 - t_air, 0.0)
    # the reference's three summary lines (cpu_engine.h:355-357 / gpu_engine.h:1251-1253), HIP-event timed
    print(f"Air update: {t_air:.6f}s, {sd.Npts * sd.Nt / 1e6 / max(t_air, 1e-12):.2f} Mvox/s")
    print(f"Boundary loop: {t_rest:.6f}s, {sd.Nb * sd.Nt / 1e6 / max(t_rest, 1e-12):.2f} Mvox/s")
    print(f"Combined (total): {el:.6f}s, {sd.Npts * sd.Nt / 1e6 / el:.2f} Mvox/s")
    sd.rescale_output()
    sd.write_outputs(a.data_dir)
    print("wrote output dataset")
    sd.print_last_samples(5)
    print(f"--Date and time: {time.ctime()}")


if __name__ == "__main__":
    main()
