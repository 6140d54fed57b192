"""GPU: the Z-slab path on ONE device -- G HIP slab engines in one process ("virtual slabs"), halo planes moved
with stream-ordered device copies through exactly the split-phase API the multi-process runner uses
(step_begin / halo tensors on the edge stream / step_end).  Must equal the single-domain oracle bit for bit.
The RCCL transport itself needs >1 GPU and is exercised by bench.py --gpus N on the multi-GPU node.
"""
import numpy as np
import pytest
import torch

import cases
import oracle
from pffdtd_amd import dist as pdist
from pffdtd_amd import slab

pytestmark = pytest.mark.gpu


def _reference(name, prec):
    sd = cases.make_sd(name, prec)
    oracle.run_sim(sd)
    return sd.u_out.copy()


@pytest.mark.parametrize("G", [2, 3])
@pytest.mark.parametrize("name,prec,variant", [("cart_outside", "single", 0), ("cart_outside", "double", 0),
                                               ("fcc2_outside", "single", 0), ("cart_lossy", "single", 3),
                                               ("fcc1_outside", "double", 7), ("fcc1_outside", "single", 3),
                                               ("fcc2_outside", "double", 3 + 256), ("cart_outside", "single", 25),
                                               ("cart_outside_oddz", "double", 25), ("fcc2_outside", "single", 4),
                                               ("fcc1_outside", "double", 4), ("cart_outside", "double", 4), ("fcc2_outside", "double", 7)])
def test_virtual_slabs_equal_single_domain(name, prec, variant, G):
    ref = _reference(name, prec)
    sd = cases.make_sd(name, prec)
    parts = [slab.split(sd, G, r) for r in range(G)]
    st = [pdist.HipSlabStepper(loc, info, 0, air_variant=variant) for loc, info in parts]
    for n in range(sd.Nt):
        for s in st:
            s.step_begin(n)
        planes = [s.halo_tensors() for s in st]
        evs = []
        for s in st:  # everything the edge streams have produced so far
            e = torch.cuda.Event()
            e.record(s.edge_stream)
            evs.append(e)
        for r in range(G - 1):
            with torch.cuda.stream(st[r + 1].edge_stream):
                st[r + 1].edge_stream.wait_event(evs[r])
                planes[r + 1][2].copy_(planes[r][1], non_blocking=True)
            with torch.cuda.stream(st[r].edge_stream):
                st[r].edge_stream.wait_event(evs[r + 1])
                planes[r][3].copy_(planes[r + 1][0], non_blocking=True)
        for s in st:
            s.step_end(n)
    for s in st:
        s.finish()
    out = slab.merge_outputs(sd, [p[0] for p in parts])
    for s in st:
        s.close()
    assert np.array_equal(out, ref), f"max|d|={np.abs(out - ref).max()}"


def test_single_rank_runner_matches():
    """world_size 1 through SlabRunner (no exchange) == engine.run."""
    ref = _reference("cart_outside", "single")
    sd = cases.make_sd("cart_outside", "single")
    runner, loc, info = pdist.make_hip_runner(sd, 0, 1, 0)
    runner.run(0, sd.Nt)
    runner.finish()
    out = pdist.gather_outputs(sd, loc, info)
    runner.st.close()
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("rank,G", [(1, 3), (0, 2), (2, 3)])
def test_a_ranks_native_rccl_exchange_moves_the_planes_a_copy_would(rank, G):
    """One process per GPU: SlabRunner.exchange through the library's own ncclSend / ncclRecv on the engine's edge stream (pf_rccl_*, a
    1-rank communicator: the rank exchanges with itself) leaves the same bits in both state grids and the same receiver samples as plain
    device copies of the same planes on the same stream -- interior and end ranks, the whole run (the physics of a rank fed its own edge
    planes is wrong by design; the data movement and its ordering are what is compared)."""
    def one(native):
        sd = cases.make_sd("cart_lossy", "single")
        runner, loc, info = pdist.make_hip_runner(sd, rank, G, 0)
        st = runner.st
        if native:
            assert runner.enable_native_rccl(0, peers=(-1 if info.first else 0, -1 if info.last else 0)), getattr(runner, "native_note", "")
            assert runner.native is not None and "native" in runner.exchange_backend
        else:
            def copies():
                s_lo, s_hi, r_lo, r_hi = st.halo_tensors()
                with st.comm_context():
                    if not info.first:
                        r_lo.copy_(s_lo, non_blocking=True)
                    if not info.last:
                        r_hi.copy_(s_hi, non_blocking=True)
            runner.exchange = copies
        rng = np.random.default_rng(7)  # a seeded field everywhere (a slab without the source would stay all zero)
        for which in (0, 1):
            g = st.eng.get_grid(which)
            st.eng.set_grid(which, (rng.uniform(-1, 1, g.shape) * 1e-3).astype(g.dtype))
        runner.run(0, sd.Nt)
        runner.finish()
        grids = [st.eng.get_grid(0).copy(), st.eng.get_grid(1).copy()]
        out = loc.u_out.copy()
        runner.close_comm()
        st.close()
        return grids, out

    g0, o0 = one(False)
    g1, o1 = one(True)
    assert np.abs(g0[1]).max() > 0
    assert np.array_equal(o0, o1)
    for a, b in zip(g0, g1):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("torch_grids", [False, True])
def test_single_domain_stepper_grids_are_the_engines(torch_grids, monkeypatch):
    """A single domain lets the engine allocate (and place) its state grids; `grids` are torch views of those allocations
    (pf_engine_state_grids): an initial field written through them is the one the engine steps.  PFFDTD_TORCH_GRIDS=1 keeps
    the torch-owned grids of the slab case; same bits either way, blocked pairs (four grids, placement sampled) included."""
    from pffdtd_amd import engine, sim_data, synth
    if torch_grids:
        monkeypatch.setenv("PFFDTD_TORCH_GRIDS", "1")
    n = (36, 64, 280)  # (the box of a blocked pair needs >= 248 columns)
    sim = synth.shoebox(*n, Nt=23, Nm=2, Mb=[11, 3], src=None, rcv=[[21, 32, 138], [6, 7, 8], [27, 54, 200]])

    def make_sd():
        s = sim_data.SimData.from_sim(sim, "single", build_mask=False)
        s.scale_input()
        return s
    sd = make_sd()
    rng = np.random.default_rng(5)
    P = engine.grid_pitch(sd.Nz, 4)
    init = [(rng.standard_normal((sd.Nx, sd.Ny, P)) * 1e-3).astype(np.float32) for _ in range(2)]
    for g in init:  # interior cells only (ghost shell and pad columns stay zero, as after any step)
        g[0], g[-1], g[:, 0], g[:, -1], g[:, :, 0], g[:, :, sd.Nz - 1:] = 0, 0, 0, 0, 0, 0
    outs = []
    for variant in (25, 40):
        sd = make_sd()
        runner, loc, info = pdist.make_hip_runner(sd, 0, 1, 0, air_variant=variant, timing=True)
        st = runner.st
        ptrs = st.eng.state_grids()
        assert ptrs[0] and ptrs[1] and ptrs[0] != ptrs[1]
        assert [g.data_ptr() for g in st.grids] == list(ptrs)
        for g, h in zip(st.grids, init):
            g.copy_(torch.from_numpy(h.reshape(sd.Nx, -1)).to(g.device))
        torch.cuda.synchronize()
        assert np.array_equal(st.eng.get_grid(0), init[0][:, :, :sd.Nz])  # the engine sees what torch wrote
        st.eng.run(0, sd.Nt)  # (as bench.py does for one GPU; SlabRunner.run would take the split-phase steps of a slab)
        st.finish()
        tm = st.eng.timing()
        if variant == 40:
            assert tm["tb2_launches"] > 0
            assert tm["place_candidates"] >= (2 if not torch_grids else 1)
            assert 0 < tm["place_ms"][1] <= tm["place_ms"][0] <= tm["place_ms"][2]
        outs.append(pdist.gather_outputs(sd, loc, info).copy())
        st.close()
    assert np.abs(outs[0]).max() > 0 and np.array_equal(outs[0], outs[1])  # single steps == blocked pairs, from a noise field


@pytest.mark.parametrize("G,Nt", [(2, 21), (3, 18)])
@pytest.mark.parametrize("src", [None, [70, 30, 150], [49, 30, 150], [45, 30, 150], [30, 33, 141]],
                         ids=["centre", "off_centre", "second_edge_plane", "second_last_edge_plane", "third_cut"])
def test_virtual_slabs_with_blocked_pairs(G, Nt, src):
    """Slab engines that own four grids step in temporally blocked pairs spanning two split-phase steps (air_variant 40
    forces it on this small cross-section); odd step counts end with a single step."""
    _run_blocked_pairs(G, Nt, src, "single")


def test_virtual_slabs_with_blocked_pairs_fp64():
    _run_blocked_pairs(2, 15, [49, 30, 150], "double")


def test_virtual_slabs_stepping_singly_choose_their_two_grids_from_the_pool():
    """Slabs whose cross-section is too small for blocked pairs (512 x 512 < 600 x 600) but whose grids are large enough to
    be worth placing (>= 64 MB): pf_engine_place_grids picks the fastest PAIR of the pool; results as the oracle's."""
    from pffdtd_amd import sim_data, synth
    n, G = (134, 512, 512), 2
    sim = synth.shoebox(*n, Nt=8, Nm=2, Mb=[11, 3], src=[66, 250, 255], rcv=[[64, 250, 250], [69, 260, 252], [7, 8, 9], [120, 500, 40]])
    ref = sim_data.SimData.from_sim(sim, "single")
    ref.scale_input()
    oracle.run_sim(ref)
    sd = sim_data.SimData.from_sim(sim, "single", build_mask=False)
    sd.scale_input()
    parts = [slab.split(sd, G, r) for r in range(G)]
    st = [pdist.HipSlabStepper(loc, info, 0, pairs=True, timing=True) for loc, info in parts]
    for s in st:
        assert not s.paired and len(s.grids) == 2 and s.grids[0].data_ptr() != s.grids[1].data_ptr()
        assert list(s.eng.state_grids()) == [g.data_ptr() for g in s.grids]
        assert s.eng.timing()["place_candidates"] >= 3
    for k in range(sd.Nt):
        for s in st:
            s.step_begin(k)
        planes = [s.halo_tensors() for s in st]
        torch.cuda.synchronize()
        planes[1][2].copy_(planes[0][1])
        planes[0][3].copy_(planes[1][0])
        torch.cuda.synchronize()
        for s in st:
            s.step_end(k)
    for s in st:
        s.finish()
    out = slab.merge_outputs(sd, [p[0] for p in parts])
    for s in st:
        s.close()
    assert np.abs(ref.u_out).max() > 0 and np.array_equal(out, ref.u_out)


def test_place_grids_argument_errors():
    from pffdtd_amd import engine, sim_data, synth
    sim = synth.shoebox(96, 64, 280, Nt=6, Nm=2, Mb=[11, 3], src=None, rcv=[[50, 30, 140]])
    sd = sim_data.SimData.from_sim(sim, "single", build_mask=False)
    sd.scale_input()
    loc, info = slab.split(sd, 2, 0)
    P = engine.grid_pitch(loc.Nz, 4)
    pool = [torch.zeros((loc.Nx, loc.Ny * P), dtype=torch.float32, device="cuda:0") for _ in range(5)]
    eng = engine.HipEngine(loc, device=0, slab_first=info.first, slab_last=info.last, x_global0=info.xlo, air_variant=40,
                           ext_u0=pool[0].data_ptr(), ext_u1=pool[1].data_ptr())
    ptrs = [g.data_ptr() for g in pool]
    with pytest.raises(RuntimeError, match="twice"):
        eng.place_grids(ptrs[:3] + [ptrs[0]])
    with pytest.raises(RuntimeError, match="at least two"):
        eng.place_grids(ptrs[:1])
    paired, idx = eng.place_grids(ptrs[:3])            # fewer than four: single steps on the first two
    assert not paired and idx == [0, 1, -1, -1]
    paired, idx = eng.place_grids(ptrs)
    assert paired and sorted(set(idx)) == sorted(idx) and all(0 <= i < 5 for i in idx)
    assert list(eng.state_grids()) == [ptrs[idx[0]], ptrs[idx[1]]]
    assert all(float(g.abs().max()) == 0.0 for g in pool)  # the timing passes leave zeros
    eng.close()
    own = engine.HipEngine(sd, device=0, air_variant=40)  # an engine with its own grids has nothing to be offered
    with pytest.raises(RuntimeError, match="own grids"):
        own.place_grids(ptrs)
    own.close()


def _run_blocked_pairs(G, Nt, src, prec):
    from pffdtd_amd import sim_data, synth
    n = (96, 64, 280)
    # (in a pair the edge stream owns two planes per side: sources / receivers in local planes 2 and Nx-3 of a slab --
    # global 46 / 49 with the cut at 48, 30..33 / 62..65 with cuts at 32 and 64 -- go through its two-plane lists)
    rcv = [[50, 30, 140], [7, 8, 9], [88, 55, 260], [46, 20, 100], [49, 40, 200], [33, 12, 40], [62, 50, 77]] + \
        ([[src[0] - 4, src[1] + 2, src[2] - 3]] if src else [])
    sim = synth.shoebox(*n, Nt=Nt, Nm=2, Mb=[11, 3], src=src, rcv=rcv)
    ref = sim_data.SimData.from_sim(sim, prec)
    ref.scale_input()
    oracle.run_sim(ref)
    sd = sim_data.SimData.from_sim(sim, prec, build_mask=False)
    sd.scale_input()
    parts = [slab.split(sd, G, r) for r in range(G)]
    st = [pdist.HipSlabStepper(loc, info, 0, pairs=True, air_variant=40, timing=True) for loc, info in parts]
    assert all(s.paired for s in st)
    for s in st:  # four distinct grids chosen from the pool the engine was offered (pf_engine_place_grids), timed there
        assert len(s.grids) in (4, 5) and len({g.data_ptr() for g in s.grids}) == len(s.grids)  # (pairs: four grids; triples, round 5: five)
        tm = s.eng.timing()
        assert tm["place_candidates"] >= 2 and 0 < tm["place_ms"][1] <= tm["place_ms"][0] <= tm["place_ms"][2]
        assert list(s.eng.state_grids()) == [s.grids[0].data_ptr(), s.grids[1].data_ptr()]
    for k in range(sd.Nt):
        for s in st:
            s.step_begin(k)
        planes = [s.halo_tensors() for s in st]
        evs = []
        for s in st:
            e = torch.cuda.Event()
            e.record(s.edge_stream)
            evs.append(e)
        for r in range(G - 1):
            with torch.cuda.stream(st[r + 1].edge_stream):
                st[r + 1].edge_stream.wait_event(evs[r])
                planes[r + 1][2].copy_(planes[r][1], non_blocking=True)
            with torch.cuda.stream(st[r].edge_stream):
                st[r].edge_stream.wait_event(evs[r + 1])
                planes[r][3].copy_(planes[r + 1][0], non_blocking=True)
        for s in st:
            s.step_end(k)
    for s in st:
        s.finish()
    assert all(s.eng.timing()["tb2_launches"] > 0 for s in st)
    out = slab.merge_outputs(sd, [p[0] for p in parts])
    for s in st:
        s.close()
    assert np.abs(ref.u_out).max() > 0
    assert np.array_equal(out, ref.u_out), f"max|d|={np.abs(out - ref.u_out).max()}"
