"""GPU: edge cases of the drop-in boundary -- ragged / empty / unsorted inputs (bit-exact vs the oracle)."""
import numpy as np
import pytest

import cases
import oracle
from pffdtd_amd import engine, sim_data, synth

pytestmark = pytest.mark.gpu


def _both(sd, **kw):
    oracle.run_sim(sd)
    ref = sd.u_out.copy()
    sd.u_out[:] = 0
    eng = engine.HipEngine(sd, **kw)
    eng.run(0, sd.Nt)
    eng.close()
    return sd.u_out.copy(), ref


@pytest.mark.parametrize("prec", ["single", "double"])
def test_unsorted_folded_lists(prec):
    """The reference's multi-GPU engine demands sorted lists (gpu_engine.h:688); this engine sorts internally."""
    sim = synth.fold_fcc(cases.make_sim("fcc1_outside"))  # folding leaves bn_ixyz unsorted
    assert (np.diff(sim["vox_out"]["bn_ixyz"]) < 0).any()
    sd = sim_data.SimData.from_sim(sim, prec)
    sd.scale_input()
    out, ref = _both(sd)
    assert np.abs(ref).max() > 0 and np.array_equal(out, ref)


def test_duplicate_and_shuffled_receivers():
    sim = cases.make_sim("cart_outside")
    c = sim["comms_out"]
    o = c["out_ixyz"]
    c["out_ixyz"] = np.concatenate([o[::-1], o[:5], o[:1]]).astype(np.int64)  # duplicates allowed (gpu_engine.h:512)
    c["Nr"] = np.int64(c["out_ixyz"].size)
    c["out_reorder"] = np.arange(c["out_ixyz"].size, dtype=np.int64)
    sd = sim_data.SimData.from_sim(sim, "single")
    sd.scale_input()
    out, ref = _both(sd, readout_chunk=7)
    assert np.array_equal(out, ref)
    n = len(o)  # rows: o reversed, then o[:5], then o[0] again -> node o[0] is read three times
    assert np.array_equal(out[n - 1], out[n]) and np.array_equal(out[n], out[-1])


def test_no_sources_no_receivers_single_step():
    sim = cases.make_sim("cart_lossy", Nt=3)
    c = sim["comms_out"]
    c["in_ixyz"] = np.zeros((0,), dtype=np.int64)
    c["in_sigs"] = np.zeros((0, 3))
    c["Ns"] = np.int64(0)
    c["out_ixyz"] = np.zeros((0,), dtype=np.int64)
    c["out_reorder"] = np.zeros((0,), dtype=np.int64)
    c["Nr"] = np.int64(0)
    sd = sim_data.SimData.from_sim(sim, "double")
    eng = engine.HipEngine(sd)
    eng.run(0, 1)
    eng.run(1, 2)
    assert not eng.get_grid(1).any()  # nothing was injected
    eng.close()


def test_step_outside_range_is_an_error():
    sd = cases.make_sd("cart_rigid", "double")
    eng = engine.HipEngine(sd)
    with pytest.raises(engine.PfError):
        eng.run(sd.Nt - 1, 2)
    eng.close()


def test_set_get_grid_roundtrip_on_padded_layout():
    sd = cases.make_sd("cart_oddz", "single")  # Nz=37: pitch 64
    eng = engine.HipEngine(sd)
    rng = np.random.default_rng(3)
    a = rng.standard_normal((sd.Nx, sd.Ny, sd.Nz)).astype(np.float32)
    eng.set_grid(0, a)
    assert np.array_equal(eng.get_grid(0), a)
    eng.close()


def test_random_field_one_step_all_paths_agree():
    """Seeded random state (every cell non-zero, so ghost flips / ABC / every neighbour weight matter) advanced a few
    steps by the oracle and by three different kernel paths."""
    sd = cases.make_sd("cart_outside_oddz", "single")
    rng = np.random.default_rng(11)
    u0 = (rng.standard_normal((sd.Nx, sd.Ny, sd.Nz)) * 1e-2).astype(np.float32)
    u1 = (rng.standard_normal((sd.Nx, sd.Ny, sd.Nz)) * 1e-2).astype(np.float32)
    e = oracle.Engine(sd)
    e.grid(0)[:] = u0
    e.grid(1)[:] = u1
    for n in range(5):
        e.step(n)
    ref = e.grid(1)[1:-1, 1:-1, 1:-1].copy()
    e.close()
    for v in (0, 3, 4, 25):
        eng = engine.HipEngine(sd, air_variant=v)
        eng.set_grid(0, u0)
        eng.set_grid(1, u1)
        eng.run(0, 5)
        got = eng.get_grid(1)[1:-1, 1:-1, 1:-1]
        eng.close()
        assert np.array_equal(got, ref), f"variant {v}: max|d|={np.abs(got - ref).max()}"


def test_long_run_ring_wraps_and_stays_bit_exact():
    """2500 steps on a 72x64x80 lossy room: the receiver ring (depth 1024) flushes three times; fp32 bits must still
    equal the CPU oracle at every sample (no drift, no race)."""
    sim = synth.shoebox(72, 64, 80, Nt=2500, Nm=3, Mb=[11, 4, 7], rigid_every=9)
    sd = sim_data.SimData.from_sim(sim, "single")
    sd.scale_input()
    out, ref = _both(sd)
    assert np.isfinite(ref).all() and np.abs(ref[:, -200:]).max() > 0
    assert np.array_equal(out, ref)


def test_graph_replay_of_the_step_loop():
    """the internal switch PF_DBG_GRAPH (0x800000): six steps per hipGraph with the step index / ring column in device counters (opt-in: no faster
    than plain launches on this stack).  Unaligned step counts, several run() calls, a ring that wraps."""
    import cases
    import oracle
    from pffdtd_amd import engine
    GRAPH = 0x800000  # PF_DBG_GRAPH: replay the step loop from a hipGraph
    for name, prec in (("cart_lossy", "single"), ("fcc2_lossy", "double"), ("cart_wall2", "single"), ("fcc1_outside", "single")):
        ref = cases.make_sd(name, prec)
        oracle.run_sim(ref)
        sd = cases.make_sd(name, prec)
        eng = engine.HipEngine(sd, readout_chunk=16, debug=GRAPH)
        eng.run(0, 3)
        eng.run(3, 25)
        eng.run(28, sd.Nt - 28)
        eng.close()
        assert np.array_equal(sd.u_out, ref.u_out), (name, prec)
