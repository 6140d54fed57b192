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
                                               ("fcc1_outside", "double", 10), ("fcc1_outside", "single", 3),
                                               ("fcc2_outside", "double", 10), ("cart_outside", "single", 20),
                                               ("cart_outside_oddz", "double", 22), ("fcc2_outside", "single", 4),
                                               ("fcc1_outside", "double", 5), ("cart_outside", "double", 6), ("fcc2_outside", "double", 7)])
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


@pytest.mark.parametrize("G,Nt", [(2, 21), (3, 18)])
@pytest.mark.parametrize("src", [None, [70, 30, 150], [49, 30, 150], [45, 30, 150], [30, 33, 141]],
                         ids=["centre", "off_centre", "second_edge_plane", "second_last_edge_plane", "third_cut"])
def test_virtual_slabs_with_blocked_pairs(G, Nt, src):
    """Slab engines that own four grids step in temporally blocked pairs spanning two split-phase steps (air_variant 40
    forces it on this small cross-section); odd step counts end with a single step."""
    _run_blocked_pairs(G, Nt, src, "single")


def test_virtual_slabs_with_blocked_pairs_fp64():
    _run_blocked_pairs(2, 15, [49, 30, 150], "double")


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
