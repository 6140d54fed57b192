"""GPU: the creation-time measurement (Engine::autotune) picks the interior kernel, its row-segment width and -- where
a box exists -- blocked pairs on the real grids.  Whatever it picks, the run must equal the CPU oracle bit for bit, and
the measurement itself (which launches every candidate into scratch) must leave the state untouched."""
import numpy as np
import pytest
import torch  # noqa: F401  (before the engine library: the first libamdhip64 loaded serves both, and torch wants its own)

import oracle
from pffdtd_amd import engine, sim_data, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,prec", [((150, 140, 309), "single"),   # 309 columns: 64 / 32 / 16-lane segments pad differently
                                    ((160, 140, 210), "double"),   # 210 doubles: 128- / 64- / 32-column segments
                                    ((96, 340, 340), "single")],   # a box exists (>= 300 cells in y and z): pairs are timed too
                         ids=["narrow_fp32", "narrow_fp64", "box_fp32"])
def test_autotuned_engine_equals_oracle(n, prec):
    Nt = 12
    rcv = [[n[0] // 2 + 3, n[1] // 2, n[2] // 2 - 2], [6, 7, 8], [n[0] - 9, n[1] - 10, n[2] - 11]]
    sim = synth.shoebox(*n, Nt=Nt, Nm=2, Mb=[11, 3], rcv=rcv)
    ref = sim_data.SimData.from_sim(sim, prec)
    ref.scale_input()
    oracle.run_sim(ref)
    assert np.abs(ref.u_out).max() > 0
    sd = sim_data.SimData.from_sim(sim, prec, build_mask=False)
    sd.scale_input()
    eng = engine.HipEngine(sd, timing=True)        # air_variant 0 = auto: the measurement runs (grid >= 2^22 cells)
    tm = eng.timing()
    assert tm["tune_ms"][0] > 0 and tm["tune_ms"][1] > 0, tm
    assert not eng.get_grid(0).any() and not eng.get_grid(1).any()   # the candidates wrote to scratch only
    eng.run(0, Nt)
    eng.close()
    assert np.array_equal(sd.u_out, ref.u_out), f"max|d|={np.abs(sd.u_out - ref.u_out).max()}"
    sd.u_out[:] = 0
    eng = engine.HipEngine(sd, debug=0x8000)        # same engine without the measurement: the static rules
    assert eng.timing()["tune_ms"][0] == 0
    eng.run(0, Nt)
    eng.close()
    assert np.array_equal(sd.u_out, ref.u_out)


def test_engine_survives_a_full_device():
    """The two spare grids of the blocked pairs and the scratch grid of the measurement are optional: with the device
    nearly full the engine must still come up (stepping singly) and produce the oracle's bits."""
    n, Nt = (160, 616, 616), 8
    sim = synth.shoebox(*n, Nt=Nt, Nm=2, Mb=[11, 3], rcv=[[83, 308, 306], [6, 7, 8]])
    ref = sim_data.SimData.from_sim(sim, "single")
    ref.scale_input()
    oracle.run_sim(ref)
    assert np.abs(ref.u_out).max() > 0
    sd = sim_data.SimData.from_sim(sim, "single", build_mask=False)
    sd.scale_input()
    eng = engine.HipEngine(sd, timing=True)          # plenty of memory: the box qualifies for pairs (may or may not win)
    eng.run(0, Nt)
    eng.close()
    assert np.array_equal(sd.u_out, ref.u_out)
    sd.u_out[:] = 0
    grid_bytes = engine.grid_pitch(n[2], 4) * n[1] * n[0] * 4
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    keep = int(3.4 * grid_bytes)                      # room for the state (2 grids) + lists + one more grid, not two
    hog = torch.empty(max(free - keep, 1), dtype=torch.uint8, device="cuda")
    try:
        eng = engine.HipEngine(sd, timing=True)
        eng.run(0, Nt)
        tm = eng.timing()
        eng.close()
    finally:
        del hog
        torch.cuda.empty_cache()
    assert tm["tb2_launches"] == 0
    assert np.array_equal(sd.u_out, ref.u_out)
