"""GPU: multi-device run_sim behind the C seam (pf_run_sim_devices / pf_run_sim, csrc/pf_multi.hip): a chain of Z-slabs,
one host thread per slab, ghost planes pulled from the neighbours with device copies on the edge stream.  A device id may
repeat, so the whole path -- slab cut in C, threads, barrier, events, copies, pairs -- runs on ONE GPU here; results must
be the single-domain CPU oracle's, bit for bit (the reference's counterpart: gpu_engine.h:516-662,739-823,993-1145)."""
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

import cases
import oracle
from pffdtd_amd import engine, h5io, sim_data, synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _ref(name, prec, **kw):
    sd = cases.make_sd(name, prec, **kw)
    oracle.run_sim(sd)
    assert np.abs(sd.u_out).max() > 0
    return sd.u_out.copy()


@pytest.mark.parametrize("name,prec", [("cart_outside", "single"), ("cart_mb11", "double"), ("fcc2_outside", "single"),
                                       ("fcc1_outside", "double"), ("cart_wall2", "single"), ("fcc2_mb11", "double")])
@pytest.mark.parametrize("G", [2, 3])
def test_virtual_slabs_through_the_c_seam(name, prec, G):
    want = _ref(name, prec)
    sd = cases.make_sd(name, prec)
    el = engine.run_sim_devices(sd, [0] * G)
    assert el > 0 and np.array_equal(sd.u_out, want)


@pytest.mark.parametrize("flags", [engine.PF_MULTI_EVEN_SPLIT, engine.PF_MULTI_ONE_THREAD,
                                   engine.PF_MULTI_EVEN_SPLIT | engine.PF_MULTI_ONE_THREAD | engine.PF_MULTI_NO_PAIRS])
def test_split_rule_and_host_threading_do_not_change_the_bits(flags):
    want = _ref("cart_outside", "single")
    sd = cases.make_sd("cart_outside", "single")
    engine.run_sim_devices(sd, [0, 0, 0, 0], multi_flags=flags)
    assert np.array_equal(sd.u_out, want)


def test_unsorted_lists_and_ring_flushes():
    """lists in arbitrary order (the reference's multi-GPU engine refuses them, gpu_engine.h:688) and a receiver ring
    much shorter than the run, so every slab flushes several times in mid-run"""
    want = _ref("cart_outside", "double")
    sd = cases.make_sd("cart_outside", "double")
    rng = np.random.default_rng(5)
    p = rng.permutation(sd.Nb)
    sd.bn_ixyz, sd.adj_bn, sd.K_bn = sd.bn_ixyz[p].copy(), sd.adj_bn[p].copy(), sd.K_bn[p].copy()
    engine.run_sim_devices(sd, [0, 0, 0], readout_chunk=7)
    assert np.array_equal(sd.u_out, want)


@pytest.mark.parametrize("prec", ["single", "double"])
def test_slabs_in_temporally_blocked_pairs(prec):
    """a box room wide enough for the two-steps-per-pass kernel, pairs forced in every slab (air_variant 40)"""
    kw = dict(Nx=100, Ny=70, Nz=276, Nt=45, wall=3, Nm=1, Mb=3, src=[47, 30, 100], rcv=[[30, 25, 96], [66, 36, 110], [48, 35, 104], [47, 4, 101]])
    sim = synth.shoebox(**kw)
    sd = sim_data.SimData.from_sim(sim, prec)
    sd.scale_input()
    oracle.run_sim(sd)
    want = sd.u_out.copy()
    assert np.abs(want).max() > 0
    for devs in ([0, 0], [0, 0, 0]):
        sd2 = sim_data.SimData.from_sim(synth.shoebox(**kw), prec)
        sd2.scale_input()
        engine.run_sim_devices(sd2, devs, multi_flags=engine.PF_MULTI_FORCE_PAIRS, air_variant=40)
        assert np.array_equal(sd2.u_out, want), devs


def test_run_sim_uses_the_device_chain_named_in_the_environment(tmp_path):
    """pf_run_sim (the seam itself) with PFFDTD_DEVICES=0,0,0 -- through the reference's own driver when it was built"""
    want = _ref("fcc2_lossy", "double")
    code = ("import sys; sys.path[:0] = [%r, %r]; import numpy as np, cases; from pffdtd_amd import engine; "
            "sd = cases.make_sd('fcc2_lossy', 'double'); engine.run_sim(sd); np.save(%r, sd.u_out)"
            % (str(ROOT), str(ROOT / "tests"), str(tmp_path / "u.npy")))
    r = subprocess.run([os.sys.executable, "-c", code], env={**os.environ, "PFFDTD_DEVICES": "0,0,0", "PFFDTD_VERBOSE": "1"},
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "3 slabs" in r.stderr
    assert np.array_equal(np.load(tmp_path / "u.npy"), want)
    exe = ROOT / "oracle" / "_ref" / "fdtd_main_hip_single.x"
    if exe.exists():
        sim = synth.sort_sim(cases.make_sim("cart_outside"))
        synth.write_folder(sim, tmp_path / "f")
        r = subprocess.run([str(exe)], cwd=tmp_path / "f", env={**os.environ, "PFFDTD_DEVICES": "0,0", "PFFDTD_VERBOSE": "1"},
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "2 slabs" in r.stderr, (r.stdout[-800:], r.stderr[-800:])
        ref = sim_data.SimData.from_folder(tmp_path / "f", "single")
        ref.scale_input()
        oracle.run_sim(ref)
        ref.rescale_output()
        assert np.array_equal(h5io.read(tmp_path / "f" / "sim_outs.h5", "u_out"), ref.u_out[ref.out_reorder, :])


def test_errors_surface_through_the_seam():
    sd = cases.make_sd("cart_rigid", "single")
    with pytest.raises(engine.PfError, match="out of range"):
        engine.run_sim_devices(sd, [0, 99])
    with pytest.raises(engine.PfError):
        engine.run_sim_devices(sd, [0] * int(sd.Nx))  # more slabs than planes (gpu_engine.h:682)


@pytest.mark.parametrize("prec", ["single", "double"])
def test_fcc_slabs_in_temporally_blocked_pairs(prec):
    """13-point folded FCC slabs with pairs forced in every slab: flips on the edge stream after the exchange, k_tb2_fcc_x over
    the slab's clean tiles, out-of-place k_air_fcc for the edge planes -- bit for bit the single-domain oracle"""
    kw = dict(Nx=100, Ny=138, Nz=276, Nt=41, fcc=True, wall=3, Nm=1, Mb=3, src=[47, 30, 101],
              rcv=[[40, 25, 97], [56, 36, 110], [48, 35, 105], [47, 6, 101]])

    def make():
        sim = synth.shoebox(**kw)
        synth.fold_fcc(sim)
        synth.sort_sim(sim)
        sd = sim_data.SimData.from_sim(sim, prec)
        sd.scale_input()
        return sd
    ref = make()
    oracle.run_sim(ref)
    assert np.abs(ref.u_out).max() > 0
    for devs in ([0, 0], [0, 0, 0]):
        sd = make()
        engine.run_sim_devices(sd, devs, multi_flags=engine.PF_MULTI_FORCE_PAIRS, air_variant=40)
        assert np.array_equal(sd.u_out, ref.u_out), devs


# ---- the chain as an object (pf_multi_*): transports, exchange self-check --------------------------------------------
@pytest.mark.parametrize("transport,flags", [(engine.PF_TRANSPORT_PEER, 0), (engine.PF_TRANSPORT_RCCL, 0),
                                             (engine.PF_TRANSPORT_RCCL, engine.PF_MULTI_ONE_THREAD)])
@pytest.mark.parametrize("name,prec", [("cart_outside", "single"), ("fcc2_mb11", "double")])
def test_chain_object_both_transports_checked_exchanges(transport, flags, name, prec):
    """pf_multi_create / _run in pieces / _destroy on three virtual slabs: ghost planes by device copies and by RCCL
    (ncclSend / ncclRecv on the edge stream; on one device every slab owns a 1-rank communicator and sends the neighbour's
    plane to itself), every exchange checksummed on both sides, receivers = the single-domain oracle's bit for bit"""
    want = _ref(name, prec)
    sd = cases.make_sd(name, prec)
    m = engine.HipMulti(sd, [0, 0, 0], multi_flags=flags, transport=transport, verify_exchange=int(sd.Nt))
    m.run(0, 7)
    m.run(7, int(sd.Nt) - 7)
    info = m.info()
    sl = [m.slab(g) for g in range(3)]
    m.close()
    assert info["nslabs"] == 3 and info["transport"] == transport
    assert ("rccl" in info["transport_name"]) == (transport == engine.PF_TRANSPORT_RCCL)
    assert info["exchange_verified"] is True and info["exchanges_checked"] == sd.Nt and info["exchange_nonzero"]
    assert sl[0]["x0"] == 0 and sl[2]["x1"] == sd.Nx and sl[0]["x1"] == sl[1]["x0"]
    assert np.array_equal(sd.u_out, want)


@pytest.mark.parametrize("transport", [engine.PF_TRANSPORT_PEER, engine.PF_TRANSPORT_RCCL])
def test_chain_object_pairs_through_both_transports(transport):
    kw = dict(Nx=100, Ny=70, Nz=276, Nt=45, wall=3, Nm=1, Mb=3, src=[47, 30, 100], rcv=[[30, 25, 96], [66, 36, 110], [48, 35, 104], [47, 4, 101]])
    sd = sim_data.SimData.from_sim(synth.shoebox(**kw), "single")
    sd.scale_input()
    oracle.run_sim(sd)
    want = sd.u_out.copy()
    sd2 = sim_data.SimData.from_sim(synth.shoebox(**kw), "single")
    sd2.scale_input()
    m = engine.HipMulti(sd2, [0, 0], multi_flags=engine.PF_MULTI_FORCE_PAIRS, air_variant=40, transport=transport, verify_exchange=45)
    m.run(0, 45)
    info, paired = m.info(), [m.slab(g)["paired"] for g in range(2)]
    m.close()
    assert all(paired) and info["exchange_verified"] is True
    assert np.array_equal(sd2.u_out, want)


@pytest.mark.parametrize("prec", ["single", "double"])
def test_slab_pairs_step_their_row_and_column_strips_as_wall_regions(prec):
    """Slabs that step in pairs hand the strips beside their box to k_wall2 (regions normal to y and z over the box's planes, on
    a stream of their own beside the box kernel; edge planes and the x walls of the end slabs stay single steps, branch state
    and node values of ALL boundary launches follow the regions' double buffers): lossy walls of two materials (11 and 3
    branches), receivers in the wall layers and next to the cuts, 2 and 3 slabs -- the oracle's bits; and wall regions did run."""
    kw = dict(Nx=112, Ny=70, Nz=276, Nt=41, wall=3, Nm=2, Mb=[11, 3], src=[55, 30, 100],  # (the source sits at the cut of the two-slab chain)
              rcv=[[30, 25, 96], [66, 36, 110], [55, 4, 104], [56, 64, 101], [57, 30, 4], [54, 31, 270], [37, 4, 4], [74, 64, 270], [4, 30, 100], [106, 40, 120]])
    sd = sim_data.SimData.from_sim(synth.shoebox(**kw), prec)
    sd.scale_input()
    oracle.run_sim(sd)
    want = sd.u_out.copy()
    assert np.abs(want).max() > 0 and np.abs(want[2:]).max() > 0
    for devs in ([0, 0], [0, 0, 0]):
        for dbg, expect in ((0, True), (0x10000000, False)):
            sd2 = sim_data.SimData.from_sim(synth.shoebox(**kw), prec)
            sd2.scale_input()
            m = engine.HipMulti(sd2, devs, multi_flags=engine.PF_MULTI_FORCE_PAIRS, air_variant=40, verify_exchange=int(sd2.Nt), debug=dbg)
            m.run(0, int(sd2.Nt))
            info = m.info()
            slabs = [m.slab(g) for g in range(len(devs))]
            blocks = [sum(sl["engine"].timing()["wall_blocks"]) for sl in slabs]
            m.close()
            assert all(sl["paired"] for sl in slabs) and info["exchange_verified"] is True
            assert all((b > 0) == expect for b in blocks), (devs, hex(dbg), blocks)
            assert np.array_equal(sd2.u_out, want), (devs, hex(dbg))


def test_chain_with_wall_regions_agrees_with_one_domain_over_many_steps():
    """700 steps of a 288 x 200 x 280 room (reflections from step ~60), too long for the oracle: ONE domain in pairs with wall regions,
    a chain of three slabs in pairs with wall regions (their own stream beside the box kernel), and the same chain with the
    single-step shell -- receivers in the wall layers and next to the cuts must agree bit for bit, every exchange checked."""
    n, ny, nz, G, K = 288, 200, 280, 3, 700
    rcv = [[n // 3, ny // 2, nz // 2 + 3], [5, 6, 7], [n - 9, ny - 10, nz // 3], [n // 2, 4, nz // 2], [n // 2 + 1, ny // 2, nz - 7],
           [n // G + 1, 9, 11], [n // G - 2, ny - 12, 13]]
    sim = synth.shoebox(n, ny, nz, Nt=K, Nm=2, Mb=[11, 3], src=[n // 2 + 7, 41, 47], rcv=rcv)
    outs = []
    for kind, dbg in (("single", 0), ("chain", 0), ("chain", 0x10000000)):
        sd = sim_data.SimData.from_sim(sim, "single", build_mask=False)
        sd.scale_input()
        if kind == "single":
            e = engine.HipEngine(sd, timing=True, air_variant=40)
            e.run(0, K)
            blocks = [sum(e.timing()["wall_blocks"])]
            e.close()
        else:
            m = engine.HipMulti(sd, [0] * G, multi_flags=engine.PF_MULTI_FORCE_PAIRS, air_variant=40, verify_exchange=K, debug=dbg)
            m.run(0, K)
            assert m.info()["exchange_verified"] is True
            blocks = [sum(m.slab(g)["engine"].timing()["wall_blocks"]) for g in range(G)]
            m.close()
        assert all((b > 0) == (dbg == 0) for b in blocks), (kind, hex(dbg), blocks)
        outs.append(sd.u_out.copy())
    assert np.abs(outs[0]).max() > 0 and np.abs(outs[0][1:]).max() > 0
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_exchange_self_check_notices_a_missing_plane(tmp_path):
    """test switch `test_drop_exchange` = 1 + n makes slab 1 skip the pull of its ghost planes in step n: the run must fail, naming the
    check -- also when the caller asked for no check at all (the fault injection switches it on)"""
    for ver in (70, 0):
        sd = cases.make_sd("cart_outside", "single")
        with pytest.raises(engine.PfError, match="self-check failed"):
            engine.run_sim_devices(sd, [0, 0, 0], verify_exchange=ver, test_drop_exchange=41)
    sd = cases.make_sd("cart_outside", "single")
    engine.run_sim_devices(sd, [0, 0, 0], verify_exchange=int(sd.Nt))


def test_transport_requests_are_validated():
    sd = cases.make_sd("cart_rigid", "single")
    with pytest.raises(engine.PfError, match="transport"):
        engine.HipMulti(sd, [0, 0], transport=7)


# ---- chains cut along FILE Z (slab engines store the x and z axes exchanged) ---------------------------------------------
@pytest.mark.parametrize("transport", [engine.PF_TRANSPORT_PEER, engine.PF_TRANSPORT_RCCL])
@pytest.mark.parametrize("name,prec", [("cart_outside", "single"), ("cart_mb11", "double"), ("fcc2_outside", "single"),
                                       ("fcc1_outside", "double"), ("fcc2_mb11", "double")])
def test_chain_cut_along_file_z(name, prec, transport):
    """PF_MULTI_CUT_Z: the slabs own ranges of FILE Z, their engines store planes of file z (unit stride along file x), ghost
    planes are contiguous storage planes -- same bits as the single-domain oracle, both transports, every exchange checked"""
    want = _ref(name, prec)
    for G in (2, 3):
        sd = cases.make_sd(name, prec)
        m = engine.HipMulti(sd, [0] * G, multi_flags=engine.PF_MULTI_CUT_Z, transport=transport, verify_exchange=int(sd.Nt))
        m.run(0, int(sd.Nt))
        info = m.info()
        lay = m.slab(0)["engine"].layout()
        zr = [(m.slab(g)["x0"], m.slab(g)["x1"]) for g in range(G)]
        m.close()
        assert info["cut_along_z"] and info["exchange_verified"] is True and info["exchange_nonzero"]
        assert lay[2] is True and lay[0][2] == sd.Nx and zr[0][0] == 0 and zr[-1][1] == sd.Nz
        assert np.array_equal(sd.u_out, want), (G, np.abs(sd.u_out - want).max())


def test_rooms_are_cut_along_file_z_by_themselves():
    """the automatic choice of the chain: a hall with six floors is cut along file z (and its engines exchange the axes), a box
    room along x like the reference's; PF_MULTI_CUT_X keeps the reference's arrangement"""
    plates = [(8, 190, 8, 90, z, z + 1) for z in (10, 18, 26, 34, 42, 50)]
    kw = dict(Nx=200, Ny=100, Nz=60, Nt=30, wall=3, Nm=2, Mb=[3, 5], blocks=plates, src=[100, 50, 6], rcv=[[104, 52, 6], [96, 47, 7], [100, 56, 5]])
    sd = sim_data.SimData.from_sim(synth.shoebox(**kw), "single")
    sd.scale_input()
    oracle.run_sim(sd)
    want = sd.u_out.copy()
    assert np.abs(want).max() > 0
    for flags, along_z in ((0, True), (engine.PF_MULTI_CUT_X, False)):
        sd2 = sim_data.SimData.from_sim(synth.shoebox(**kw), "single")
        sd2.scale_input()
        m = engine.HipMulti(sd2, [0, 0], multi_flags=flags, verify_exchange=30)
        m.run(0, 30)
        info = m.info()
        m.close()
        assert info["cut_along_z"] == along_z and info["exchange_verified"] is True
        assert np.array_equal(sd2.u_out, want), flags


def test_run_sim_runs_a_room_as_two_slabs_on_one_device(tmp_path):
    """pf_run_sim with ONE device in use: a room (cut along file z) is stepped as two slabs on that device -- their kernels overlap;
    PFFDTD_DEVICES=0 (a chain of one named device) keeps one domain; same bits either way"""
    plates = [(8, 190, 8, 90, z, z + 1) for z in (10, 18, 26, 34, 42, 50, 58, 66)]
    kw = dict(Nx=200, Ny=100, Nz=76, Nt=24, wall=3, Nm=2, Mb=[3, 5], blocks=plates, src=[100, 50, 6], rcv=[[104, 52, 6], [96, 47, 7], [100, 56, 5]])
    sd = sim_data.SimData.from_sim(synth.shoebox(**kw), "single")
    sd.scale_input()
    oracle.run_sim(sd)
    want = sd.u_out.copy()
    assert np.abs(want).max() > 0
    code = ("import sys; sys.path[:0] = [%r]; import numpy as np; from pffdtd_amd import engine, sim_data, synth; "
            "sd = sim_data.SimData.from_sim(synth.shoebox(**%r), 'single'); sd.scale_input(); engine.run_sim(sd); np.save(%r, sd.u_out)"
            % (str(ROOT), kw, str(tmp_path / "u.npy")))
    for extra, expect in (({"PFFDTD_NGPUS": "1"}, "2 slabs cut along file z"), ({"PFFDTD_DEVICES": "0"}, None)):
        r = subprocess.run([os.sys.executable, "-c", code], env={**os.environ, "PFFDTD_VERBOSE": "1", **extra},
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert (expect in r.stderr) if expect else ("slabs" not in r.stderr), r.stderr[-1500:]
        assert np.array_equal(np.load(tmp_path / "u.npy"), want)


# ---- round 5: first contact with a multi-device box cannot end without a run ---------------------------------------------
@pytest.mark.parametrize("flags", [0, engine.PF_MULTI_ONE_THREAD])
@pytest.mark.parametrize("name,prec", [("cart_outside", "single"), ("fcc2_mb11", "double")])
def test_host_staged_transport_gives_the_oracles_bits(name, prec, flags):
    """PF_TRANSPORT_HOST, the last resort: edge planes -> pinned bounce buffer on the sender's edge stream, -> ghost planes on the
    receiver's, the two sides meeting on the host; three virtual slabs, every exchange checksummed on both sides."""
    want = _ref(name, prec)
    sd = cases.make_sd(name, prec)
    m = engine.HipMulti(sd, [0, 0, 0], multi_flags=flags, transport=engine.PF_TRANSPORT_HOST, verify_exchange=int(sd.Nt))
    m.run(0, 5)
    m.run(5, int(sd.Nt) - 5)
    info = m.info()
    m.close()
    assert info["transport"] == engine.PF_TRANSPORT_HOST and info["transport_name"] == "host-staged"
    assert info["exchange_verified"] is True and info["exchanges_checked"] == sd.Nt and info["exchange_nonzero"]
    assert np.array_equal(sd.u_out, want)


def test_host_staged_transport_carries_slab_pairs():
    kw = dict(Nx=100, Ny=70, Nz=276, Nt=45, wall=3, Nm=1, Mb=3, src=[47, 30, 100], rcv=[[30, 25, 96], [66, 36, 110], [48, 35, 104], [47, 4, 101]])
    sd = sim_data.SimData.from_sim(synth.shoebox(**kw), "single")
    sd.scale_input()
    oracle.run_sim(sd)
    want = sd.u_out.copy()
    sd2 = sim_data.SimData.from_sim(synth.shoebox(**kw), "single")
    sd2.scale_input()
    m = engine.HipMulti(sd2, [0, 0, 0], multi_flags=engine.PF_MULTI_FORCE_PAIRS, air_variant=40, transport=engine.PF_TRANSPORT_HOST, verify_exchange=45)
    m.run(0, 45)
    info, paired = m.info(), [m.slab(g)["paired"] for g in range(3)]
    m.close()
    assert all(paired) and info["exchange_verified"] is True
    assert np.array_equal(sd2.u_out, want)


@pytest.mark.parametrize("faults,expect", [(1, engine.PF_TRANSPORT_RCCL), (3, engine.PF_TRANSPORT_HOST)], ids=["no_peer_access", "no_peer_access_no_rccl"])
def test_automatic_transport_falls_back_edge_by_edge(faults, expect):
    """the test switch `test_faults` (csrc/pf_debug.h) forces every fallback edge of PF_TRANSPORT_AUTO on one device: without peer access the chain takes RCCL,
    without RCCL as well the host-staged copies -- and says why (pf_multi_info.transport_note); the bits stay the oracle's.  An
    explicitly requested transport that is unavailable remains an error."""
    want = _ref("cart_outside", "single")
    sd = cases.make_sd("cart_outside", "single")
    m = engine.HipMulti(sd, [0, 0, 0], transport=engine.PF_TRANSPORT_AUTO, verify_exchange=int(sd.Nt), test_faults=faults)
    m.run(0, int(sd.Nt))
    info = m.info()
    m.close()
    assert info["transport"] == expect, info
    assert "cannot access each other" in info["transport_note"]
    if expect == engine.PF_TRANSPORT_HOST:
        assert "RCCL" in info["transport_note"]
    assert info["exchange_verified"] is True and np.array_equal(sd.u_out, want)
    sd = cases.make_sd("cart_outside", "single")
    with pytest.raises(engine.PfError, match="cannot access each other"):
        engine.HipMulti(sd, [0, 0], transport=engine.PF_TRANSPORT_PEER, test_faults=1)
    with pytest.raises(engine.PfError, match="RCCL"):
        engine.HipMulti(sd, [0, 0], transport=engine.PF_TRANSPORT_RCCL, test_faults=2)


def test_a_hung_slab_thread_becomes_an_error_not_a_hang(tmp_path):
    """the test switch `test_faults` (csrc/pf_debug.h) = 4: the host thread of slab 1 stalls before the barrier of its fourth step.  The watchdog of the other
    slabs' barrier waits (PFFDTD_BARRIER_TIMEOUT_S) turns that into pf_last_error instead of a process that never returns.  In a
    process of its own: the chain object is abandoned with its stuck thread."""
    code = ("import sys, time; sys.path[:0] = [%r, %r]; import cases; from pffdtd_amd import engine; "
            "sd = cases.make_sd('cart_outside', 'single'); m = engine.HipMulti(sd, [0, 0, 0], test_faults=4); t0 = time.time()\n"
            "try:\n    m.run(0, int(sd.Nt)); print('NO ERROR', flush=True)\n"
            "except engine.PfError as e:\n    print('ERROR after %%.1f s: %%s' %% (time.time() - t0, e), flush=True)\n"
            "import os; os._exit(0)\n" % (str(ROOT), str(ROOT / "tests")))
    r = subprocess.run([os.sys.executable, "-c", code], env={**os.environ, "PFFDTD_BARRIER_TIMEOUT_S": "2"}, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ERROR after" in r.stdout and "hung" in r.stdout, (r.stdout, r.stderr[-1500:])


# ---- round 5: slabs in TRIPLES across three split-phase steps ---------------------------------------------------------------
@pytest.mark.parametrize("prec", ["single", "double"])
@pytest.mark.parametrize("transport", [engine.PF_TRANSPORT_PEER, engine.PF_TRANSPORT_RCCL])
def test_slabs_step_in_triples_across_three_split_phase_steps(prec, transport):
    """A slab offered five grids whose wall regions fit steps THREE steps per pass (k_tb3 over its box, three edge planes per side on the
    edge stream, wall regions for two steps and one single step for the third, ghost planes exchanged after every step): lossy walls of
    two materials, receivers in the wall layers, in the box and next to the cuts, 2 and 3 slabs, step counts that leave one and two
    single steps over -- the oracle's bits, every exchange checked; PF_MULTI_NO_TRIPLES keeps the round-4 pairs."""
    nz = 276 if prec == "single" else 264
    kw = dict(Nx=124, Ny=70, Nz=nz, Nt=41, wall=3, Nm=2, Mb=[11, 3], src=[61, 30, 100],
              rcv=[[30, 25, 96], [70, 36, 110], [61, 4, 104], [62, 63, 101], [63, 30, 4], [60, 31, nz - 7], [41, 4, 4], [82, 63, nz - 7], [4, 30, 100], [117, 40, 120], [4, 4, 4], [118, 64, nz - 7], [118, 20, 50], [5, 33, 90]])
    sd = sim_data.SimData.from_sim(synth.shoebox(**kw), prec)
    sd.scale_input()
    oracle.run_sim(sd)
    want = sd.u_out.copy()
    assert np.abs(want).max() > 0 and np.abs(want[2:]).max() > 0
    # round 6: a slab in triples takes its shell's three steps in the FIRST split-phase step (three-step y / z regions + bricks for the four bars
    # along x) unless a source sits within reach of them (two slabs: the source is at the cut) -- debug 0x400000: the round-5 shell
    # 0x40 (PF_DBG_SRC_TILES_SINGLE): the source's and the receivers' tiles step singly, the sources by k_io, as until late in round 6
    # (default: the sources inside k_tb3_src, launched first on the regions' second stream; a receiver's tile stores its u^{n+1})
    for devs, flags, spp, dbg in (([0, 0], 0, 3, 0), ([0, 0, 0], 0, 3, 0), ([0, 0, 0], 0, 3, 0x400000), ([0, 0, 0], 0, 3, 0x40), ([0, 0], engine.PF_MULTI_NO_TRIPLES, 2, 0)):
        sd2 = sim_data.SimData.from_sim(synth.shoebox(**kw), prec)
        sd2.scale_input()
        m = engine.HipMulti(sd2, devs, multi_flags=engine.PF_MULTI_FORCE_PAIRS | flags, air_variant=40, transport=transport, verify_exchange=int(sd2.Nt), timing=1, debug=dbg)
        m.run(0, 20)
        m.run(20, int(sd2.Nt) - 20)
        info = m.info()
        tms = [m.slab(g)["engine"].timing() for g in range(len(devs))]
        m.close()
        assert info["exchange_verified"] is True
        assert all(t["tb_steps_per_pass"] == spp and t["tb2_launches"] > 0 and sum(t["wall_blocks"]) > 0 for t in tms), (devs, flags, [t["tb_steps_per_pass"] for t in tms])
        if len(devs) == 3 and spp == 3:  # (fp64: two steps + one; three-step tables are fp32)
            want3 = prec == "single" and dbg in (0, 0x40)
            # the slab that holds the source: no tile steps singly unless asked to
            assert (max(t["tb2_dirty_tiles"] for t in tms) > 0) == (dbg == 0x40), (hex(dbg), [t["tb2_dirty_tiles"] for t in tms])
            # (bits 0x10 / 0x20: an end slab's own x wall is a region's and the bricks' too -- no single steps of its planes)
            assert all((t["wall_three_steps"] == (9 | (0x10 if g == 0 else 0) | (0x20 if g == 2 else 0)) and t["wall_bricks"] > 0) == want3 for g, t in enumerate(tms)), \
                (hex(dbg), [(t["wall_three_steps"], t["wall_bricks"]) for t in tms])
        assert np.array_equal(sd2.u_out, want), (devs, flags, hex(dbg))


# ---- round 5: the wall planes' weights of the balanced cut are measured when the chain is created ---------------------------
def test_partition_weights_are_measured_at_creation_and_change_no_bits():
    """PF_MULTI_MEASURE_WEIGHTS: pf_multi_create on a scene with at least 2^24 cells per slab and an interior rank (three slabs or more) times three one-rank cost models (an interior
    rank, one with half as many planes again, the first rank with its wall) and scales the compiled-in wall-plane weights by what it finds
    (pf_multi_info.wall_scale, wall_measured).  The calibration chains write receiver rows like any chain: what sd.u_out held must be
    back afterwards.  Whatever the cut, the chain's receivers equal one domain's bit for bit; a caller-fixed factor skips the measurement,
    and without the flag the compiled-in weights cut the chain."""
    n, ny, nz, K, G = 480, 330, 322, 66, 3
    rcv = [[n // 3 + 3, 8, 10], [n // 3 - 1, 5, 6], [n // 3 - 4, 14, 18], [n // 3 + 12, 4, 4], [n // 3, 20, 9]]  # (both sides of the cut, some in the wall layers)
    sim = synth.shoebox(n, ny, nz, Nt=K, Nm=2, Mb=[11, 3], src=[n // 3 + 7, 12, 14], rcv=rcv)
    sd = sim_data.SimData.from_sim(sim, "single", build_mask=False)
    sd.scale_input()
    e = engine.HipEngine(sd)
    e.run(0, K)
    e.flush_outputs()
    e.close()
    want = sd.u_out.copy()
    assert np.abs(want).max() > 0 and all(np.abs(w).max() > 0 for w in want)
    seen = {}
    for key, flags, kw in (("measured", engine.PF_MULTI_MEASURE_WEIGHTS, {}), ("fixed", 0, {}), ("given", engine.PF_MULTI_MEASURE_WEIGHTS, dict(wall_scale=1.7))):
        sd2 = sim_data.SimData.from_sim(sim, "single", build_mask=False)
        sd2.scale_input()
        sd2.u_out[:] = 7.25
        m = engine.HipMulti(sd2, [0] * G, multi_flags=flags, verify_exchange=K, **kw)
        assert np.all(sd2.u_out == 7.25), key
        info = m.info()
        m.run(0, K)
        assert m.info()["exchange_verified"] is True
        seen[key] = (info["wall_scale"], info["wall_measured"], [m.slab(g)["x1"] - m.slab(g)["x0"] for g in range(G)])
        m.close()
        assert np.array_equal(sd2.u_out, want), key
    assert seen["measured"][1] is True and 0.25 <= seen["measured"][0] <= 4.0, seen
    assert seen["fixed"][:2] == (1.0, False) and seen["given"][:2] == (1.7, False), seen
    assert seen["measured"][2] == [x1 - x0 for x0, x1 in engine.slab_partition(sd, G, wall_scale=seen["measured"][0])], seen
    print("wall weights:", seen)
