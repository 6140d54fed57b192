"""GPU: `SimEngine.gather_slice` (the live-slice feed of python/fdtd/sim_fdtd.py:630-658) against the CPU
oracle's field: after n steps every plane the host asks for -- x, y and z cuts -- equals the oracle's u1 bit for bit on its
interior (the ghost rim of a slice is only refreshed at the start of the next step in the reference, cpu_engine.h:145-172,
while the HIP engine materialises it on request: not compared); on the FCC checkerboard grid the hole fill of
nb_fcc_fill_plot_holes (:888-895) is applied on top of identical data."""
import numpy as np
import pytest

import cases
import oracle
from pffdtd_amd import sim_fdtd, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,prec", [("cart_outside", "double"), ("cart_mb11", "single"), ("fcc2_outside", "double"), ("fcc1_outside", "double")])
def test_slices_equal_the_oracle_field(tmp_path, name, prec):
    sim = cases.make_sim(name)
    synth.write_folder(sim, tmp_path)
    se = sim_fdtd.SimEngine(tmp_path, precision=prec)
    se.load_h5_data(); se.setup_mask(); se.allocate_mem(); se.set_coeffs(); se.checks()
    ref_sd = cases.make_sd(name, prec, scale=False)
    ref = oracle.Engine(ref_sd)
    nsteps = 23
    se.run_steps(0, nsteps)
    for n in range(nsteps):
        ref.step(n)
    u1 = ref.grid(1)
    assert np.abs(u1).max() > 0
    Nx, Ny, Nz = ref_sd.Nx, ref_sd.Ny, ref_sd.Nz
    for kw, want in ((dict(ix=Nx // 2), u1[Nx // 2]), (dict(ix=1), u1[1]), (dict(iy=Ny // 3), u1[:, Ny // 3]),
                     (dict(iy=Ny - 2), u1[:, Ny - 2]), (dict(iz=Nz // 2 + 1), u1[:, :, Nz // 2 + 1]), (dict(iz=1), u1[:, :, 1])):
        got = se.gather_slice(**kw)
        want = np.array(want)
        if ref_sd.fcc_flag == 1:  # the reference fills the non-existent checkerboard cells for plotting: same rule on the oracle's data
            k = list(kw.values())[0]
            i1, i2 = np.meshgrid(np.arange(1, want.shape[0] - 1), np.arange(1, want.shape[1] - 1), indexing="ij")
            holes = ((i1 + i2 + k) % 2) == 1
            avg = 0.25 * (want[2:, 1:-1] + want[:-2, 1:-1] + want[1:-1, 2:] + want[1:-1, :-2])
            want[1:-1, 1:-1][holes] = avg[holes]
        assert got.shape == want.shape and np.array_equal(got[2:-2, 2:-2], want[2:-2, 2:-2]), kw
    ref.close()
