"""Python host with the interface of the reference's `SimEngine` (python/fdtd/sim_fdtd.py:38-697), running
the time loop on the MI355X through the C ABI instead of numba.

    python -m pffdtd_amd.sim_fdtd --data_dir <sim_data folder> [--nsteps k] [--precision double|single] [--gpu i]

Same method names and call order as the reference's `main()` (sim_fdtd.py:898-937): load_h5_data -> setup_mask ->
allocate_mem -> set_coeffs -> checks -> run_all -> save_outputs -> print_last_samples.  Like the reference Python
engine it does not rescale the input (`scale_input` belongs to the C flow: see pffdtd_amd/fdtd_main.py).
Plotting (`--plot`) is out of scope.  `--energy` runs the reference's energy-conservation diagnostic on the device.
"""
import argparse
import time
from pathlib import Path

from . import engine, sim_data


class SimEngine:
    def __init__(self, data_dir, energy_on=False, nthreads=None, precision="double", device=0):
        self.data_dir = Path(data_dir)
        self.energy_on = energy_on
        self.precision = precision
        self.device = device
        if energy_on and precision != "double":
            raise ValueError("the energy diagnostic runs in double precision (like the reference Python engine)")
        self.print(f"HIP engine: {engine.lib().pf_version().decode()}, {engine.device_count()} device(s)")

    def print(self, fstring):
        print(f"--ENGINE: {fstring}")

    def load_h5_data(self):
        self.print("loading data..")
        sd = sim_data.SimData.from_folder(self.data_dir, self.precision)
        self.sd = sd
        for k in ("Nx", "Ny", "Nz", "Nt", "Ns", "Nr", "Nb", "Nbl", "Nba", "l", "l2", "Ts", "fcc_flag", "in_sigs",
                  "in_ixyz", "out_ixyz", "out_reorder", "bn_ixyz", "adj_bn", "bnl_ixyz", "mat_bnl", "Mb"):
            setattr(self, k, getattr(sd, k))
        self.fcc = sd.fcc_flag > 0
        self.print(f"Nx={sd.Nx} Ny={sd.Ny} Nz={sd.Nz}")
        self.print(f"l={sd.l} l2={sd.l2} fcc={self.fcc}")
        self.print(f"Nr={sd.Nr} Ns={sd.Ns} Nt={sd.Nt}")

    def setup_mask(self):
        self.print("setting up bn mask..")  # built on the device by the engine (pf_kernels.h: k_mask_init/k_mask_set)

    def allocate_mem(self):
        self.print("allocating mem..")
        self.eng = engine.HipEngine(self.sd, device=self.device, energy=self.energy_on)
        self.u_out = self.sd.u_out
        if self.energy_on:  # sim_fdtd.py:181-185
            import numpy as np
            sd = self.sd
            if sd.h is None or sd.c is None:
                raise ValueError("sim_consts.h5 must hold h and c for the energy diagnostic")
            self.H_tot = np.zeros((sd.Nt,), dtype=np.float64)
            self.E_lost = np.zeros((sd.Nt + 1,), dtype=np.float64)
            self.E_in = np.zeros((sd.Nt + 1,), dtype=np.float64)
            self.eng.energy_cfg(sd.h, sd.c, sd.Ts, sd.DEF)

    def set_coeffs(self):
        pass  # coefficients are derived by the loader (fdtd_data.h:186-194,441-457)

    def checks(self):
        sd = self.sd
        assert (sd.saf_bnl <= (12 if self.fcc else 6)).all()  # sim_fdtd.py:286-291

    def run_steps(self, nstart, nsteps):
        if self.energy_on:
            self.eng.run_energy(nstart, nsteps, self.H_tot, self.E_lost, self.E_in)
        else:
            self.eng.run(nstart, nsteps)

    def run_all(self, nsteps=0):
        """nsteps: steps per engine call (the reference's batch size for progress / plotting); 0 = the whole run in one
        call, which lets the engine keep its receiver ring and step-pair schedule uninterrupted."""
        self.print("running..")
        sd = self.sd
        t0 = time.perf_counter()
        nsteps = int(nsteps) if int(nsteps) > 0 else max(int(sd.Nt), 1)
        for n in range(0, sd.Nt, nsteps):
            self.run_steps(n, min(nsteps, sd.Nt - n))
        self.eng.sync()
        t = time.perf_counter() - t0
        self.print(f"Run-time loop: {t:.6f}, {sd.Nt * sd.Npts / 1e6 / t:.2f} MVox/s")

    def gather_slice(self, ix=None, iy=None, iz=None):
        """One plane of the current field u1 (sim_fdtd.py:630-658); the reference fills the checkerboard holes of an FCC
        subgrid for plotting (nb_fcc_fill_plot_holes, :888-895), reproduced with numpy."""
        import numpy as np
        u1 = self.eng.get_grid(1)
        if ix is not None:
            sl, k = u1[ix, :, :].copy(), ix
        elif iy is not None:
            sl, k = u1[:, iy, :].copy(), iy
        else:
            sl, k = u1[:, :, iz].copy(), iz
        if self.fcc and self.sd.fcc_flag == 1:
            i1, i2 = np.meshgrid(np.arange(1, sl.shape[0] - 1), np.arange(1, sl.shape[1] - 1), indexing="ij")
            holes = ((i1 + i2 + k) % 2) == 1
            avg = 0.25 * (sl[2:, 1:-1] + sl[:-2, 1:-1] + sl[1:-1, 2:] + sl[1:-1, :-2])
            inner = sl[1:-1, 1:-1]
            inner[holes] = avg[holes]
        return sl

    def save_outputs(self):
        self.sd.write_outputs(self.data_dir)
        self.print(f"saved outputs in {self.data_dir}")

    def print_last_samples(self, Np):
        self.print("GRID OUTPUTS")
        sd = self.sd
        for i in range(sd.Nr):
            self.print(f"out {i}")
            for n in range(max(sd.Nt - Np, 0), sd.Nt):
                self.print(f"sample {n}: {sd.u_out[sd.out_reorder[i], n]:.16e}")

    def print_last_energy(self, Np):
        self.print("ENERGY")
        for n in range(max(self.sd.Nt - Np, 0), self.sd.Nt):
            self.print(f"normalised energy balance:{rel_diff(self.H_tot[n] + self.E_lost[n], self.E_in[n]):.16e}")


def rel_diff(x0, x1):
    """(x0-x1)/2^floor(log2(x0)) -- python/common/myfuncs.py:164-165 (0 where x0 is not positive)."""
    import numpy as np
    x0, x1 = np.asarray(x0, dtype=np.float64), np.asarray(x1, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (x0 - x1) / (2.0 ** np.floor(np.log2(x0)))
    r = np.where(x0 > 0, r, 0.0)
    return r if r.ndim else float(r)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--data_dir", type=str, required=True, help="run directory")
    p.add_argument("--nsteps", type=int, default=0, help="run in batches of steps (0 = all at once)")
    p.add_argument("--nthreads", type=int, default=None, help="ignored (kept for CLI compatibility)")
    p.add_argument("--energy", action="store_true", help="do energy calc")
    p.add_argument("--plot", action="store_true", help="not supported (visualisation is out of scope)")
    p.add_argument("--json_model", type=str, default=None, help="ignored (only the reference's --plot uses it)")
    p.add_argument("--draw_backend", type=str, default="mayavi", help="ignored (only the reference's --plot uses it)")
    p.add_argument("--abc", action="store_true", help="unused, as in the reference")
    p.add_argument("--precision", default="double", choices=["double", "single"])
    p.add_argument("--gpu", type=int, default=0)
    a = p.parse_args()
    if a.plot:
        raise SystemExit("--plot is not supported")
    eng = SimEngine(a.data_dir, energy_on=a.energy, nthreads=a.nthreads, precision=a.precision, device=a.gpu)
    eng.load_h5_data()
    eng.setup_mask()
    eng.allocate_mem()
    eng.set_coeffs()
    eng.checks()
    eng.run_all(a.nsteps)
    eng.save_outputs()
    eng.print_last_samples(5)
    if a.energy:
        eng.print_last_energy(5)


if __name__ == "__main__":
    main()
