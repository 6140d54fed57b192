"""GPU: barrier-free 7-point kernel with the rigid boundary update done in-kernel from the cell-byte grid
(pf_kernels.h RIGB, engine flag debug 0x800 / auto rule): bit-identical to the oracle."""
import numpy as np
import pytest

import cases
import oracle
from pffdtd_amd import engine, scenes, sim_data

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["single", "double"])
@pytest.mark.parametrize("name", ["cart_lossy", "cart_rigid", "cart_mb11", "cart_oddz", "cart_outside", "cart_outside_oddz"])
def test_in_kernel_rigid_update_matches_oracle(name, prec):
    ref = cases.make_sd(name, prec)
    oracle.run_sim(ref)
    for variant in (0, 4):
        sd = cases.make_sd(name, prec)
        eng = engine.HipEngine(sd, air_variant=variant, debug=0x800)
        eng.run(0, sd.Nt)
        g1 = eng.get_grid(1)
        eng.close()
        assert np.array_equal(sd.u_out, ref.u_out), (name, prec, variant)
        base = cases.make_sd(name, prec)
        e2 = engine.HipEngine(base, air_variant=variant, debug=0x1000)
        e2.run(0, base.Nt)
        assert np.array_equal(e2.get_grid(1)[1:-1, 1:-1, 1:-1], g1[1:-1, 1:-1, 1:-1])
        e2.close()


def test_in_kernel_rigid_update_on_the_ctk_church(tmp_path):
    """Real geometry (chairs, tilted surfaces): scene export -> sim folder -> both boundary paths -> identical receivers."""
    from pffdtd_amd.sim_setup import sim_setup
    mats = scenes.write_materials(tmp_path / "materials")
    folder = tmp_path / "gpu"
    sim_setup(**scenes.setup_kwargs("ctk_cart_gpu", folder, mats, save_folder_gpu=folder, compress=0, duration=0.02, PPW=6.0, fmax=400.0))
    outs = []
    for dbg in (0x800, 0x1000):
        sd = sim_data.SimData.from_folder(folder, "single", build_mask=False)
        sd.scale_input()
        eng = engine.HipEngine(sd, debug=dbg)
        eng.run(0, sd.Nt)
        eng.close()
        outs.append(sd.u_out.copy())
    ref = sim_data.SimData.from_folder(folder, "single")
    ref.scale_input()
    oracle.run_sim(ref)
    assert np.abs(ref.u_out).max() > 0
    assert np.array_equal(outs[0], ref.u_out) and np.array_equal(outs[1], ref.u_out)
