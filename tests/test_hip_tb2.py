"""GPU: temporal blocking (two steps per pass over the boundary-free box, pf_tb2.h / Engine::step_pair) must leave
every bit where the single-step engine and the CPU oracle put it.  Grids here are the smallest that qualify
(the box needs >= 248 columns), with step counts that mix quads of blocked steps, single steps and ring flushes."""
import numpy as np
import pytest

import oracle
from pffdtd_amd import engine, sim_data, synth

pytestmark = pytest.mark.gpu


def scene(src, Nt=23, n=(36, 64, 280), **kw):
    w = kw.get("wall", 3) + 3  # receivers stay clear of the wall layers
    rcv = [[n[0] // 2 + 3, n[1] // 2, n[2] // 2 - 2], [w, w + 1, w + 2], [n[0] - w - 3, n[1] - w - 4, 200]]
    if src is not None and src[0] < kw.get("wall", 3):
        rcv.append([src[0], src[1] + 9, src[2] + 11])  # a receiver on the source's side of the wall (the box is closed)
    return synth.shoebox(*n, Nt=Nt, Nm=2, Mb=[11, 3], src=src, rcv=rcv, **kw)


def run(sim, variant, prec="single", **kw):
    sd = sim_data.SimData.from_sim(sim, prec, build_mask=False)
    sd.scale_input()
    eng = engine.HipEngine(sd, air_variant=variant, timing=True, **kw)
    eng.run(0, sd.Nt)
    tm = eng.timing()
    g = [eng.get_grid(0).copy(), eng.get_grid(1).copy()]
    eng.close()
    return sd.u_out.copy(), g, tm


@pytest.mark.parametrize("src,kw", [(None, {}), ([3, 30, 140], dict(n=(36, 72, 280), wall=6)), ([18, 8, 12], {})],
                         ids=["centre", "outside_wall", "near_corner"])
def test_blocked_steps_match_single_steps_and_oracle(src, kw):
    sim = scene(src, **kw)
    ref = sim_data.SimData.from_sim(sim, "single")
    ref.scale_input()
    oracle.run_sim(ref)
    assert np.abs(ref.u_out).max() > 0
    base_out, base_g, tm0 = run(sim, 25)            # plain single-step lean kernel
    assert np.array_equal(base_out, ref.u_out) and tm0["tb2_launches"] == 0
    for variant, chunk in ((41, 0), (40, 0), (0, 0), (40, 7)):
        out, g, tm = run(sim, variant, readout_chunk=chunk)
        assert np.array_equal(out, ref.u_out), (variant, chunk)
        for a, b in zip(g, base_g):                   # whole fields, interior (ghost shell is materialised on request)
            assert np.abs(b).max() > 0
            assert np.array_equal(a[1:-1, 1:-1, 1:-1], b[1:-1, 1:-1, 1:-1]), (variant, chunk)
        if variant == 40:  # (auto engages only when the box holds >= 60 % of the grid: not at this size)
            assert tm["tb2_launches"] > 0 and tm["tb2_cells"] > 0.1 * 36 * 64 * 280, (variant, tm)
        nx, ny, nz = (int(sim["vox_out"][k]) for k in ("Nx", "Ny", "Nz"))
        assert tm["steps"] == sim["comms_out"]["Nt"]


def test_a_stray_boundary_node_only_dirties_its_tiles():
    sim = scene(None, n=(36, 64, 280), rigid_every=0)
    # one rigid node floating in the middle of the room: round 1 had no box for such a room; now only the tiles around it
    # step singly
    from pffdtd_amd import synth as sy  # noqa: F401
    sd = sim_data.SimData.from_sim(sim, "single", build_mask=False)
    mid = (18 * 64 + 32) * 280 + 140
    if mid not in set(sd.bn_ixyz.tolist()):
        v = sim["vox_out"]
        order = np.argsort(np.append(v["bn_ixyz"], mid))
        v["bn_ixyz"] = np.append(v["bn_ixyz"], mid)[order]
        v["adj_bn"] = np.vstack([v["adj_bn"], np.zeros((1, v["adj_bn"].shape[1]), dtype=v["adj_bn"].dtype)])[order]
        v["mat_bn"] = np.append(v["mat_bn"], -1)[order].astype(v["mat_bn"].dtype)
        v["saf_bn"] = np.append(v["saf_bn"], 6.0)[order]
        v["Nb"] = np.int64(v["bn_ixyz"].size)
    ref = sim_data.SimData.from_sim(sim, "single")
    ref.scale_input()
    oracle.run_sim(ref)
    out, _, tm = run(sim, 0)
    assert tm["tb2_launches"] == 0  # (auto: the cross-section is far too small for pairs to pay)
    assert np.array_equal(out, ref.u_out)
    out, _, tm = run(sim, 40)
    full = (36 - 10) * (64 - 10) * (272 - 8)
    assert tm["tb2_launches"] > 0 and 0.3 * full < tm["tb2_cells"] < full, tm
    assert np.array_equal(out, ref.u_out)


BLOCKS = [(12, 17, 20, 40, 60, 130), (22, 24, 8, 12, 150, 260), (8, 9, 44, 52, 20, 30)]


@pytest.mark.parametrize("prec", ["single", "double"])
@pytest.mark.parametrize("dbg", [0, 0x100, 0x200], ids=["lw64", "lw32", "lw16"])
def test_rooms_with_interior_geometry_block_tile_by_tile(prec, dbg):
    """Solid blocks standing in the room (pillar, balcony, step): the tiles of the box that hold their surface nodes --
    or the source -- take single steps (k_tb1_tile), all others pairs (k_tb2_reg over the clean-tile list), with row
    segments of 64 / 32 / 16 lanes.  Whole fields and receivers equal the single-step engine's and the oracle's."""
    n = (36, 96, 280)
    rcv = [[20, 50, 140], [15, 41, 133], [24, 49, 141], [19, 40, 131], [14, 50, 146]]  # (the front moves 0.58 cells a step)
    sim = synth.shoebox(*n, Nt=45, Nm=2, Mb=[11, 3], src=[19, 45, 138], rcv=rcv, blocks=BLOCKS)
    ref = sim_data.SimData.from_sim(sim, prec)
    ref.scale_input()
    oracle.run_sim(ref)
    assert (np.abs(ref.u_out).max(axis=1) > 0).all()
    _, base_g, _ = run(sim, 25, prec=prec)
    for variant, chunk in ((40, 0), (40, 5)):
        out, g, tm = run(sim, variant, prec=prec, readout_chunk=chunk, debug=dbg)
        assert tm["tb2_launches"] > 0 and 0 < tm["tb2_cells"] < 0.9 * n[0] * n[1] * n[2], tm
        assert np.array_equal(out, ref.u_out), (variant, chunk)
        for a, b in zip(g, base_g):
            assert np.array_equal(a[1:-1, 1:-1, 1:-1], b[1:-1, 1:-1, 1:-1]), (variant, chunk)


@pytest.mark.parametrize("n,wall", [((37, 67, 283), 3), ((41, 75, 291), 5), ((34, 62, 270), 4), ((36, 64, 325), 3), ((36, 64, 571), 3)],
                         ids=["odd", "deep_walls", "tight", "sliver_60_columns", "two_tiles_and_sliver"])
def test_blocked_steps_on_awkward_sizes(n, wall):
    """Row counts that are not multiples of the strip tile height, column counts that are not multiples of 4, pitch
    padding next to the right column strip, wall depths that move the box."""
    sim = scene([n[0] // 2, n[1] // 2 - 3, n[2] // 2 + 5], Nt=14, n=n, wall=wall)
    ref = sim_data.SimData.from_sim(sim, "single")
    ref.scale_input()
    oracle.run_sim(ref)
    out, g, tm = run(sim, 40)
    assert tm["tb2_launches"] > 0
    assert np.array_equal(out, ref.u_out)
    _, base_g, _ = run(sim, 25)
    for a, b in zip(g, base_g):
        assert np.array_equal(a[1:-1, 1:-1, 1:-1], b[1:-1, 1:-1, 1:-1])


def test_blocked_steps_long_run():
    """300 steps (75 quads) with reflections off the lossy walls back into the box; receiver ring of 64 wraps 4 times."""
    sim = scene(None, Nt=301, n=(36, 64, 280))
    ref = sim_data.SimData.from_sim(sim, "single")
    ref.scale_input()
    oracle.run_sim(ref)
    out, _, tm = run(sim, 40, readout_chunk=64)
    assert tm["tb2_launches"] >= 95 and tm["steps"] == 301  # (one launch per pair or triple: the source only dirties its own tiles)
    assert np.array_equal(out, ref.u_out)


def test_strip_kernel_boundary_modes_give_the_same_bits():
    """Boundary nodes inside the column strips: by default the strip kernel does their rigid update and k_fd_sel the branch
    ODEs (dense); debug 0x20000000 = neither (the list kernel visits every node, the round-1 arrangement and the fallback).
    Same bits as the oracle in both, with a source in a corner so that every wall is live."""
    sim = scene([18, 8, 12], Nt=18)
    ref = sim_data.SimData.from_sim(sim, "single")
    ref.scale_input()
    oracle.run_sim(ref)
    for dbg in (0, 0x20000000):
        out, _, tm = run(sim, 40, debug=dbg)
        assert tm["tb2_launches"] > 0 and np.array_equal(out, ref.u_out), hex(dbg)


@pytest.mark.parametrize("src,kw", [(None, {}), ([3, 30, 140], dict(n=(36, 72, 280), wall=6)), ([18, 8, 12], dict(n=(37, 67, 283))),
                                    (None, dict(n=(36, 64, 294)))],
                         ids=["centre", "outside_wall", "odd", "sliver_30_columns"])
def test_blocked_steps_in_double_precision(src, kw):
    """fp64: 128 columns per wave (two doubles per lane), otherwise the same kernels."""
    sim = scene(src, Nt=31, **kw)
    ref = sim_data.SimData.from_sim(sim, "double")
    ref.scale_input()
    oracle.run_sim(ref)
    assert np.abs(ref.u_out).max() > 0
    _, base_g, _ = run(sim, 25, prec="double")
    for variant, dbg in ((41, 0), (40, 0), (40, 0x20000000)):
        out, g, tm = run(sim, variant, prec="double", readout_chunk=8, debug=dbg)
        assert np.array_equal(out, ref.u_out), (variant, dbg)
        for a, b in zip(g, base_g):
            assert np.array_equal(a[1:-1, 1:-1, 1:-1], b[1:-1, 1:-1, 1:-1]), (variant, dbg)
        if variant == 40:
            assert tm["tb2_launches"] > 0


def fcc_scene(n=(36, 70, 280), Nt=25, src=None, blocks=(), wall=3, rcv=None):
    """folded FCC room with a stored grid of n (unfolded Ny = 2 (n[1] - 1)); receivers near the source"""
    Nyu = 2 * (n[1] - 1)
    src = src or [n[0] // 2, n[1] // 2, n[2] // 2]
    src = [src[0], src[1], src[2] + (sum(src) % 2)]  # an existing (even) node of the subgrid
    rcv = rcv or [[src[0] + dx, src[1] + dy, src[2] + dz + ((dx + dy + dz) % 2)] for dx, dy, dz in ((2, 3, -4), (-5, 2, 6), (3, -6, 9), (-2, -3, -8))]
    sim = synth.shoebox(n[0], Nyu, n[2], Nt=Nt, fcc=True, Nm=2, Mb=[11, 3], src=src, rcv=rcv, blocks=blocks, wall=wall)
    synth.fold_fcc(sim)
    synth.sort_sim(sim)
    return sim


@pytest.mark.parametrize("prec", ["single", "double"])
@pytest.mark.parametrize("dbg", [0, 0x100, 0x200], ids=["lw64", "lw32", "lw16"])
@pytest.mark.parametrize("blocks", [(), ((12, 17, 20, 40, 60, 130), (22, 24, 8, 12, 150, 260))], ids=["box", "blocks"])
def test_fcc_blocked_pairs_match_single_steps_and_oracle(prec, dbg, blocks):
    """13-point folded FCC: k_tb2_fcc over the clean tiles, k_air_fcc tiles / k_zstrip_fcc / whole-plane slabs around them,
    ghost flips in memory -- receivers equal the oracle's, whole fields the single-step engine's."""
    sim = fcc_scene(blocks=blocks, src=[19, 45, 138] if blocks else None)
    ref = sim_data.SimData.from_sim(sim, prec)
    ref.scale_input()
    oracle.run_sim(ref)
    assert np.abs(ref.u_out).max() > 0
    base_out, base_g, tm0 = run(sim, 0, prec=prec, debug=0x4000)   # single steps
    assert tm0["tb2_launches"] == 0 and np.array_equal(base_out, ref.u_out)
    for chunk in (0, 6):
        out, g, tm = run(sim, 40, prec=prec, readout_chunk=chunk, debug=dbg)
        assert tm["tb2_launches"] > 0 and tm["tb2_cells"] > 0, tm
        assert np.array_equal(out, ref.u_out), chunk
        for a, b in zip(g, base_g):
            assert np.array_equal(a[1:-1, 1:-1, 1:-1], b[1:-1, 1:-1, 1:-1]), chunk


def test_fcc_blocked_pairs_with_live_abc_and_odd_sizes():
    """source outside the room (the wave runs into the ABC shell and the ghost flips on every face), sizes that are no
    multiples of the tile dimensions"""
    sim = fcc_scene(n=(38, 67, 286), Nt=31, src=[3, 30, 140], wall=7, rcv=[[2, 33, 137], [4, 25, 151], [3, 30, 128], [33, 4, 200]])
    ref = sim_data.SimData.from_sim(sim, "single")
    ref.scale_input()
    oracle.run_sim(ref)
    assert np.abs(ref.u_out).max() > 0
    out, g, tm = run(sim, 40)
    assert tm["tb2_launches"] > 0 and np.array_equal(out, ref.u_out)
    _, base_g, _ = run(sim, 0, debug=0x4000)
    for a, b in zip(g, base_g):
        assert np.array_equal(a[1:-1, 1:-1, 1:-1], b[1:-1, 1:-1, 1:-1])


# ---- wall regions (pf_wall.h): the shell of a blocked pair in pairs too --------------------------------------------------
WALL_MODES = [(0, "default"), (0x400000, "generic_blocks"), (0x8000000, "all_generic"), (0x4000000, "one_stream"), (0x2000000, "split_strips"), (0x10000000, "single_step_shell")]


@pytest.mark.parametrize("prec", ["single", "double"])
@pytest.mark.parametrize("dbg,label", WALL_MODES, ids=[m[1] for m in WALL_MODES])
def test_wall_regions_give_the_oracles_bits(prec, dbg, label):
    """k_wall2 steps the wall layers, ABC cells and ghost mirrors around the box twice per pass (branch state double-buffered,
    blocks of alike pencils decoded in scalar registers, the others generically): receivers equal the oracle's, whole fields
    the single-step engine's -- three materials mixed along the walls, 11 / 3 / 7 branches (generic mode: every 13th lossy node
    rigid, so that no two neighbouring pencils look alike)."""
    n = (38, 66, 280)
    sim = synth.shoebox(*n, Nt=63, Nm=3, Mb=[11, 3, 7], rigid_every=13 if dbg == 0x8000000 else 0, src=[19, 30, 140],
                        rcv=[[19, 33, 150], [5, 29, 141], [32, 31, 139], [18, 5, 142], [20, 60, 138]])  # next to four walls
    ref = sim_data.SimData.from_sim(sim, prec)
    ref.scale_input()
    oracle.run_sim(ref)
    assert (np.abs(ref.u_out[::8]).max(axis=1) > 0).all()
    _, base_g, _ = run(sim, 25, prec=prec)
    for chunk in (0, 6):
        out, g, tm = run(sim, 40, prec=prec, readout_chunk=chunk, debug=dbg)
        assert tm["tb2_launches"] > 0 and tm["steps"] == 63
        if dbg == 0x10000000:
            assert tm["wall_blocks"] == [0, 0]
        elif dbg == 0x8000000:
            assert tm["wall_blocks"][0] == 0 and tm["wall_blocks"][1] > 0
        elif dbg == 0x400000:  # the round-5 arrangement: edges and corners as generic blocks of k_wall2
            assert tm["wall_blocks"][0] > 0 and tm["wall_blocks"][1] > 0 and tm["wall_bricks"] == 0, tm
        else:  # the frame as bricks (pf_brick.h): what is left of the regions is alike
            assert tm["wall_blocks"][0] > 0 and tm["wall_bricks"] > 0, tm
        assert np.array_equal(out, ref.u_out), (label, chunk)
        for a, b in zip(g, base_g):
            assert np.array_equal(a[1:-1, 1:-1, 1:-1], b[1:-1, 1:-1, 1:-1]), (label, chunk)


@pytest.mark.parametrize("n,wall,expect", [((37, 67, 280), 3, True), ((41, 75, 280), 4, True), ((36, 64, 276), 3, True), ((40, 70, 528), 4, True),
                                           ((37, 67, 283), 3, False), ((36, 64, 325), 3, False), ((41, 75, 280), 5, False)],
                         ids=["odd", "deep_walls", "narrow_sliver", "two_tiles", "strips_too_wide", "sliver_60_columns", "walls_too_deep"])
@pytest.mark.parametrize("numerics", [engine.PF_NUM_CPU_EXACT, engine.PF_NUM_GPU_SAFEGUARDED], ids=["exact", "safeguarded"])
def test_wall_regions_from_random_fields(n, wall, expect, numerics):
    """Every cell live from step 0 (seeded random u^{n-1}, u^n): ghost mirrors on all three axes, ABC faces / edges / corners,
    both wall layers on every face, odd sizes.  Pairs with wall regions against the single-step engine, all cells."""
    sim = scene([n[0] // 2, n[1] // 2 - 3, n[2] // 2 + 5], Nt=10, n=n, wall=wall)
    rng = np.random.default_rng(17)
    init = [(rng.standard_normal(n) * 1e-2).astype(np.float32) for _ in range(2)]
    fields = {}
    for variant, dbg in ((25, 0), (40, 0), (40, 0x400000), (40, 0x8000000)):  # (bricks; generic blocks for the frame; for everything)
        sd = sim_data.SimData.from_sim(sim, "single", build_mask=False)
        sd.scale_input()
        eng = engine.HipEngine(sd, air_variant=variant, timing=True, debug=dbg, numerics=numerics)
        for k in (0, 1):
            eng.set_grid(k, init[k])
        eng.run(0, sd.Nt)
        tm = eng.timing()
        fields[(variant, dbg)] = (sd.u_out.copy(), eng.get_grid(0).copy(), eng.get_grid(1).copy())
        eng.close()
        if variant == 40:  # (column strips wider than the pencils -- a sliver of more than 16 columns -- keep the single-step shell)
            assert (sum(tm["wall_blocks"]) > 0) == expect, tm["wall_blocks"]
    base = fields[(25, 0)]
    for key, f in fields.items():
        assert np.array_equal(f[0], base[0]), key
        for a, b in zip(f[1:], base[1:]):
            assert np.array_equal(a[1:-1, 1:-1, 1:-1], b[1:-1, 1:-1, 1:-1]), key


@pytest.mark.parametrize("mb", [[4, 2], [1], [12, 5]], ids=["Mb4_2", "Mb1", "Mb12_5"])
@pytest.mark.parametrize("prec", ["single", "double"])
def test_wall_regions_with_few_and_many_branches(mb, prec):
    """k_wall2 moves 4 or 12 branch-state slots per node (a compile-time bound of the scene's largest branch count): scenes whose
    materials have up to 4, exactly 1 and the maximum of 12 branches, alike and generic blocks, both precisions -- the oracle's
    receivers and the single-step engine's fields, every cell live from step 0."""
    n = (37, 67, 280)
    rcv = [[n[0] // 2 + 3, n[1] // 2, n[2] // 2 - 2], [6, 7, 8], [n[0] - 9, n[1] - 10, 200], [5, 30, 140], [18, 6, 270]]
    sim = synth.shoebox(*n, Nt=12, Nm=len(mb), Mb=mb, src=[n[0] // 2, n[1] // 2 - 3, n[2] // 2 + 5], rcv=rcv, wall=3)
    dt = np.float32 if prec == "single" else np.float64
    rng = np.random.default_rng(23)
    init = [(rng.standard_normal(n) * 1e-2).astype(dt) for _ in range(2)]
    ref = sim_data.SimData.from_sim(sim, prec)
    ref.scale_input()
    e = oracle.Engine(ref)
    for k in (0, 1):
        e.grid(k)[...] = init[k]
    for i in range(ref.Nt):
        e.step(i)
    ref_u1 = e.grid(1).copy()
    e.close()
    for dbg in (0, 0x8000000):
        sd = sim_data.SimData.from_sim(sim, prec, build_mask=False)
        sd.scale_input()
        eng = engine.HipEngine(sd, air_variant=40, timing=True, debug=dbg)
        for k in (0, 1):
            eng.set_grid(k, init[k])
        eng.run(0, sd.Nt)
        tm, u1 = eng.timing(), eng.get_grid(1)
        eng.close()
        assert tm["tb2_launches"] > 0 and sum(tm["wall_blocks"]) > 0, (hex(dbg), tm["wall_blocks"])
        assert np.array_equal(sd.u_out, ref.u_out), hex(dbg)
        assert np.array_equal(u1[1:-1, 1:-1, 1:-1], ref_u1[1:-1, 1:-1, 1:-1]), hex(dbg)


def test_wall_regions_step_aside_for_a_source_in_the_shell():
    """A source within a cell of the shell is added between the two steps, which a region that keeps u^{n+1} in registers cannot
    see: such scenes keep the single-step shell (and the oracle's bits)."""
    sim = scene([18, 8, 12], Nt=12)
    ref = sim_data.SimData.from_sim(sim, "single")
    ref.scale_input()
    oracle.run_sim(ref)
    out, _, tm = run(sim, 40)
    assert tm["tb2_launches"] > 0 and tm["wall_blocks"] == [0, 0]
    assert np.array_equal(out, ref.u_out)


@pytest.mark.parametrize("prec", ["single", "double"])
@pytest.mark.parametrize("src,kw", [(None, {}), ([18, 8, 12], dict(n=(37, 67, 283)))], ids=["centre", "corner_odd"])
def test_blocked_pairs_in_the_gpu_safeguarded_arithmetic(prec, src, kw):
    """PF_NUM_GPU_SAFEGUARDED in the pair kernels (round 4: k_tb2_reg, k_tb1_tile, k_air_zstrip, k_wall2 with the towards-zero pairwise sums
    and the two FMAs of gpu_engine.h:220-242, 288-314): forced pairs give the bits of the oracle's restatement of that
    arithmetic, and differ from the CPU-exact mode only in the roundings."""
    sim = scene(src, Nt=31, **kw)
    sd = sim_data.SimData.from_sim(sim, prec)
    sd.scale_input()
    e = oracle.Engine(sd, safeguarded=True)
    for n in range(sd.Nt):
        e.step(n)
    ref_out, ref_u1 = sd.u_out.copy(), e.grid(1).copy()
    e.close()
    exact = sim_data.SimData.from_sim(sim, prec)
    exact.scale_input()
    oracle.run_sim(exact)
    for variant in (41, 40):
        out, g, tm = run(sim, variant, prec=prec, numerics=engine.PF_NUM_GPU_SAFEGUARDED)
        assert np.array_equal(out, ref_out), variant
        assert np.array_equal(g[1][1:-1, 1:-1, 1:-1], ref_u1[1:-1, 1:-1, 1:-1]), variant
        if variant == 40:
            assert tm["tb2_launches"] > 0
            assert (sum(tm["wall_blocks"]) > 0) == (src is None)  # (wall regions in the safeguarded arithmetic too, k_wall2<..., SG>; not with a source in the shell)
    peak = np.abs(exact.u_out).max()
    d = np.abs(ref_out - exact.u_out).max()
    assert 0 < d <= (3e-5 if prec == "single" else 1e-12) * peak


# ---- round 5: 13-point pairs in the safeguarded arithmetic, pairs on grids stored with exchanged axes -----------------------
@pytest.mark.parametrize("prec", ["single", "double"])
@pytest.mark.parametrize("dbg", [0x400, 0x100, 0x200], ids=["lw64", "lw32", "lw16"])
def test_fcc_blocked_pairs_in_the_gpu_safeguarded_arithmetic(prec, dbg):
    """PF_NUM_GPU_SAFEGUARDED in the 13-point pair kernels (k_tb2_fcc_w<..., SG> with 64-lane segments, k_tb2_fcc<..., SG> with
    32 / 16) and their single-step shell (k_zstrip_fcc, k_air_fcc tiles): the towards-zero tree of gpu_engine.h:257-267 and the two
    FMAs of :268 -- forced pairs give the bits of the oracle's restatement of that arithmetic, receivers and whole field."""
    sim = fcc_scene(blocks=((12, 17, 20, 40, 60, 130),), src=[19, 45, 138])
    sd = sim_data.SimData.from_sim(sim, prec)
    sd.scale_input()
    e = oracle.Engine(sd, safeguarded=True)
    for n in range(sd.Nt):
        e.step(n)
    ref_out, ref_u1 = sd.u_out.copy(), e.grid(1).copy()
    e.close()
    assert np.abs(ref_out).max() > 0
    exact = sim_data.SimData.from_sim(sim, prec)
    exact.scale_input()
    oracle.run_sim(exact)
    assert not np.array_equal(ref_out, exact.u_out)  # (the two arithmetics do differ in the last bits)
    for variant, d in ((0, 0x4000), (40, dbg)):
        out, g, tm = run(sim, variant, prec=prec, numerics=engine.PF_NUM_GPU_SAFEGUARDED, debug=d)
        assert (tm["tb2_launches"] > 0) == (variant == 40), (variant, tm)
        if variant == 40:
            assert tm["tb2_lw"] == {0x400: 64, 0x100: 32, 0x200: 16}[dbg]
        assert np.array_equal(out, ref_out), variant
        assert np.array_equal(g[1][1:-1, 1:-1, 1:-1], ref_u1[1:-1, 1:-1, 1:-1]), variant


@pytest.mark.parametrize("prec", ["single", "double"])
def test_fcc_pair_kernels_old_and_new_give_the_same_bits(prec):
    """k_tb2_fcc_w (every product a2 * u rounded once per cell and shared by the twelve sums it enters) against k_tb2_fcc_x
    (debug 0x40000: the products inside every sum) and the oracle."""
    sim = fcc_scene(Nt=21)
    ref = sim_data.SimData.from_sim(sim, prec)
    ref.scale_input()
    oracle.run_sim(ref)
    fields = []
    for dbg in (0x400, 0x400 | 0x40000):
        out, g, tm = run(sim, 40, prec=prec, debug=dbg)
        assert tm["tb2_launches"] > 0 and tm["tb2_lw"] == 64
        assert np.array_equal(out, ref.u_out), hex(dbg)
        fields.append(g)
    for a, b in zip(*fields):
        assert np.array_equal(a[1:-1, 1:-1, 1:-1], b[1:-1, 1:-1, 1:-1])


@pytest.mark.parametrize("prec", ["single", "double"])
@pytest.mark.parametrize("numerics", [engine.PF_NUM_CPU_EXACT, engine.PF_NUM_GPU_SAFEGUARDED], ids=["exact", "safeguarded"])
@pytest.mark.parametrize("fcc", [False, True], ids=["cart", "fcc"])
def test_blocked_pairs_on_a_grid_stored_with_exchanged_axes(fcc, numerics, prec):
    """Rooms are stored with the file's x and z axes exchanged (unit stride along the longest axis, debug 0x1000 forces it): the
    pair kernels then work in storage coordinates but take their neighbours in the FILE's order (k_tb2_reg / k_tb1_tile /
    k_air_zstrip / k_tb2_fcc_w / k_zstrip_fcc<..., SWZ>) -- forced pairs with a block standing in the room give the oracle's bits."""
    blocks = ((60, 130, 20, 40, 12, 17), (150, 260, 8, 12, 22, 24))
    if fcc:
        n = (280, 70, 36)
        src = [138, 45, 19]
        src = [src[0], src[1], src[2] + (sum(src) % 2)]
        rcv = [[src[0] + dx, src[1] + dy, src[2] + dz + ((dx + dy + dz) % 2)] for dx, dy, dz in ((-4, 3, 2), (6, 2, -5), (9, -6, 3), (-8, -3, -2))]
        sim = synth.shoebox(n[0], 2 * (n[1] - 1), n[2], Nt=33, fcc=True, Nm=2, Mb=[11, 3], src=src, rcv=rcv, blocks=blocks)
        synth.fold_fcc(sim)
        synth.sort_sim(sim)
    else:
        n = (280, 64, 36)
        sim = synth.shoebox(*n, Nt=33, Nm=2, Mb=[11, 3], src=[138, 45, 19], rcv=[[140, 50, 20], [133, 41, 15], [141, 49, 24], [131, 40, 19]], blocks=blocks)
    sd = sim_data.SimData.from_sim(sim, prec)
    sd.scale_input()
    e = oracle.Engine(sd, safeguarded=numerics == engine.PF_NUM_GPU_SAFEGUARDED)
    for k in range(sd.Nt):
        e.step(k)
    ref_out, ref_u1 = sd.u_out.copy(), e.grid(1).copy()
    e.close()
    assert (np.abs(ref_out).max(axis=1) > 0).all()
    for variant, dbg in ((0, 0x1000 | 0x4000), (40, 0x1000)):
        out, g, tm = run(sim, variant, prec=prec, numerics=numerics, debug=dbg, readout_chunk=7)
        assert (tm["tb2_launches"] > 0) == (variant == 40), (variant, tm)
        assert np.array_equal(out, ref_out), variant
        assert np.array_equal(g[1][1:-1, 1:-1, 1:-1], ref_u1[1:-1, 1:-1, 1:-1]), variant


def test_pairs_with_wall_regions_report_their_air_time():
    """pf_timing.air_ms_total (the CLI's 'Air update' line and the --progress fields) with wall regions on: the alike blocks' launches
    and the box kernel are recorded -- not 0, not more than the steps took."""
    sim = scene(None, Nt=41, n=(38, 66, 280))
    out, _, tm = run(sim, 40)
    assert tm["tb2_launches"] > 0 and sum(tm["wall_blocks"]) > 0
    assert tm["steps"] == 41 and tm["air_launches"] > 0
    assert 0.2 * tm["step_ms_total"] < tm["air_ms_total"] <= tm["step_ms_total"], tm


# ---- round 5: three steps per pass (k_tb3, Engine::step_triple) ---------------------------------------------------------------
def triple_scene(Nt=40, n=(48, 100, 280), wall=3, **kw):
    """a box room whose 20-row tiles stay partly clean: the source dirties the middle tile, one receiver the last one, the other
    receivers sit in the wall layers (the shell) and next to the source (receivers are the 8 corner nodes of a cell: p .. p+1)"""
    src = [n[0] // 2, n[1] // 2, n[2] // 2]
    lo, hi = wall + 1, [d - wall - 3 for d in n]
    rcv = [[src[0] + 2, src[1] - 1, src[2] + 3], [lo, src[1] - 3, src[2] + 2], [hi[0], src[1] + 2, src[2] - 5], [src[0], hi[1], src[2] + 4],
           [src[0] + 3, n[1] - 12, src[2] - 3], [lo + 1, lo, src[2] + 1], [src[0] - 2, src[1] + 1, hi[2]]]
    return synth.shoebox(*n, Nt=Nt, Nm=2, Mb=[11, 3], src=src, rcv=rcv, wall=wall, **kw)


@pytest.mark.parametrize("prec", ["single", "double"])
@pytest.mark.parametrize("numerics", [engine.PF_NUM_CPU_EXACT, engine.PF_NUM_GPU_SAFEGUARDED], ids=["exact", "safeguarded"])
def test_three_steps_per_pass_give_the_oracles_bits(prec, numerics):
    """k_tb3 advances the box by three steps per pass (u^{n+1} never stored); the shell takes its steps as wall regions and bricks (or two
    steps + one).  Round 6: the tiles within two cells of the source run k_tb3<..., SRC> (the samples added in registers after every
    stage), a receiver's tile merely stores its u^{n+1}: no tile steps singly; PF_DBG_SRC_TILES_SINGLE (0x40): as until then, the source's
    and a receiver's tiles three single steps, reading the u^{n+1} their flagged neighbours left behind.  Receivers in the box, in the
    shell and in the wall layers equal the oracle's; whole fields the single-step engine's.  Step counts that mix triples, single
    steps and ring flushes."""
    sim = triple_scene(Nt=100, n=(48, 100, 280 if prec == "single" else 264))  # (column counts whose strips fit the wall regions' pencils)
    sd = sim_data.SimData.from_sim(sim, prec)
    sd.scale_input()
    e = oracle.Engine(sd, safeguarded=numerics == engine.PF_NUM_GPU_SAFEGUARDED)
    for k in range(sd.Nt):
        e.step(k)
    ref_out, ref_u1 = sd.u_out.copy(), e.grid(1).copy()
    e.close()
    assert (np.abs(ref_out[:-8]).max(axis=1) > 0).all()  # (all but the receiver in the far z wall layer: the random-field test covers that)
    base_out, base_g, _ = run(sim, 25, prec=prec, numerics=numerics)
    assert np.array_equal(base_out, ref_out)
    # (0x80000: the third step by the list kernels instead of the regions' one-step form; 0x400000: the frame as generic blocks of k_wall2 -- round 5 -- instead of bricks)
    # 0x1000000 / 0x40000000: the x / y regions / the column strips two steps + one instead of three in one pass (k_wall2<..., NS = 3>)
    for variant, chunk, dbg in ((40, 0, 0), (40, 7, 0), (40, 8, 0x4000000), (40, 0, 0x80000), (40, 0, 0x400000), (40, 5, 0x1000000), (40, 0, 0x40000000), (40, 0, 0x800), (40, 0, 0x40), (40, 7, 0x400040)):  # (0x800: the three-step bodies with run-time pencil geometry)
        out, g, tm = run(sim, variant, prec=prec, numerics=numerics, readout_chunk=chunk, debug=dbg)
        assert tm["tb_steps_per_pass"] == 3 and tm["tb2_launches"] > 0 and sum(tm["wall_blocks"]) > 0, (variant, chunk, tm)
        assert (tm["wall_bricks"] > 0) == (dbg & 0x480000 == 0), (hex(dbg), tm)
        # fp32: the whole shell in one pass with the box kernel (x / y regions: bit 0, column strips of 20-cell pencils: bit 3); fp64: two steps + one
        want3 = 0 if (prec == "double" or dbg & 0x1480000) else (1 if dbg & 0x40000000 else 9)
        assert tm["wall_three_steps"] == want3, (hex(dbg), tm)
        assert (tm["tb2_dirty_tiles"] >= 2 if dbg & 0x40 else tm["tb2_dirty_tiles"] == 0) and tm["steps"] == 100, (hex(dbg), tm)
        assert np.array_equal(out, ref_out), (variant, chunk, hex(dbg))
        assert np.array_equal(g[1][1:-1, 1:-1, 1:-1], ref_u1[1:-1, 1:-1, 1:-1]), (variant, chunk)
        for a, b in zip(g, base_g):
            assert np.array_equal(a[1:-1, 1:-1, 1:-1], b[1:-1, 1:-1, 1:-1]), (variant, chunk)
    out, _, tm = run(sim, 40, prec=prec, numerics=numerics, debug=0x20000)  # never triples: the round-4 pairs
    assert tm["tb_steps_per_pass"] == 2 and np.array_equal(out, ref_out)


@pytest.mark.parametrize("prec", ["single", "double"])
@pytest.mark.parametrize("sx,sy,sz", [(24, 24, 140), (24, 25, 140), (24, 26, 140), (24, 27, 140), (18, 50, 140), (19, 50, 140), (20, 50, 140),
                                      (31, 44, 251), (32, 45, 254), (33, 46, 257), (24, 86, 20)])
def test_sources_added_inside_k_tb3_at_tile_borders(sx, sy, sz, prec):
    """The eight corner nodes of the source cell (p .. p + 1 on every axis) straddle the borders of k_tb3's tiles -- 20 rows, 248 (fp64:
    120) core columns, x chunks of 13 planes -- in turn: every tile that computes one of those cells, in its core or in the two cells of
    halo a later stage still sees, must add the same samples after the same stage.  One node listed twice (the reference's loop is
    serial: both samples land, in list order).  Receivers beside the source and far from it equal the oracle's, whole fields too."""
    n = (48, 100, 280 if prec == "single" else 264)
    sz = min(sz, n[2] - 12)
    rcv = [[sx + 2, sy - 1, sz + 3], [sx - 3, sy + 2, sz - 2], [4, 50, 100], [30, 90, 200]]
    sim = synth.shoebox(*n, Nt=43, Nm=2, Mb=[11, 3], src=[sx, sy, sz], rcv=rcv, wall=3)

    def make(mask=False):
        sd = sim_data.SimData.from_sim(sim, prec, build_mask=mask)
        k = 3  # one of the eight nodes a second time, with a signal of its own
        sd.in_ixyz = np.ascontiguousarray(np.concatenate([sd.in_ixyz, sd.in_ixyz[k:k + 1]]))
        sd.in_sigs = np.ascontiguousarray(np.concatenate([sd.in_sigs, 0.37 * np.roll(sd.in_sigs[k:k + 1], 5, axis=1)]))
        sd.Ns += 1
        sd.scale_input()
        return sd

    ref = make(True)
    e = oracle.Engine(ref)
    for k in range(ref.Nt):
        e.step(k)
    ref_u1 = e.grid(1).copy()
    e.close()
    assert np.abs(ref.u_out).max() > 0
    for dbg in (0, 0x40):
        sd = make()
        eng = engine.HipEngine(sd, air_variant=40, timing=True, debug=dbg)
        eng.run(0, sd.Nt)
        tm = eng.timing()
        g1 = eng.get_grid(1).copy()
        eng.close()
        assert tm["tb_steps_per_pass"] == 3 and tm["tb2_launches"] > 0, tm
        assert (tm["tb2_dirty_tiles"] > 0) == bool(dbg), (hex(dbg), tm)
        assert np.array_equal(sd.u_out, ref.u_out), hex(dbg)
        assert np.array_equal(g1[1:-1, 1:-1, 1:-1], ref_u1[1:-1, 1:-1, 1:-1]), hex(dbg)


@pytest.mark.parametrize("n,wall,triples", [((47, 101, 280), 3, True), ((50, 96, 280), 4, False), ((44, 90, 528), 3, True), ((47, 101, 283), 3, False)],
                         ids=["odd", "deep_walls", "two_tiles", "strips_too_wide"])
def test_three_steps_per_pass_from_random_fields(n, wall, triples):
    """every cell live from step 0 (seeded random u^{n-1}, u^n): ghost mirrors, ABC faces / edges / corners, both wall layers on
    every face, odd sizes, a second column tile -- triples against the single-step engine, all cells, 11 steps (3 triples + a pair
    on the triples' tiles).  Walls four cells deep or strips too wide for the pencils: no wall regions for the triples' box, the
    engine falls back to pairs (same bits)."""
    sim = triple_scene(Nt=11, n=n, wall=wall)
    rng = np.random.default_rng(29)
    init = [(rng.standard_normal(n) * 1e-2).astype(np.float32) for _ in range(2)]
    fields = {}
    for variant, dbg in ((25, 0), (40, 0), (40, 0x400000), (40, 0x1000000), (40, 0x800)):  # (0x400000: the frame as generic blocks instead of bricks; 0x1000000: regions two steps + one; 0x800: run-time pencil geometry)
        sd = sim_data.SimData.from_sim(sim, "single", build_mask=False)
        sd.scale_input()
        eng = engine.HipEngine(sd, air_variant=variant, timing=True, debug=dbg)
        for k in (0, 1):
            eng.set_grid(k, init[k])
        eng.run(0, sd.Nt)
        tm = eng.timing()
        fields[(variant, dbg)] = (sd.u_out.copy(), eng.get_grid(0).copy(), eng.get_grid(1).copy())
        eng.close()
        if variant == 40:
            assert tm["tb_steps_per_pass"] == (3 if triples else 2) and tm["tb2_launches"] == 4, tm  # (3 triples + a pair; pairs come in twos: 2 x 2 + 3 single steps)
            assert (tm["wall_bricks"] > 0) == (sum(tm["wall_blocks"]) > 0 and dbg & 0x400000 == 0), tm
    for key in ((40, 0), (40, 0x400000), (40, 0x1000000), (40, 0x800)):
        for a, b in zip(fields[key], fields[(25, 0)]):
            assert np.array_equal(a if a.ndim == 2 else a[1:-1, 1:-1, 1:-1], b if b.ndim == 2 else b[1:-1, 1:-1, 1:-1]), key


def test_three_step_regions_fall_back_where_pencils_are_not_alike():
    """Every 13th frequency-dependent node rigid: hardly a block of the wall regions has pencils that all look alike, so the regions cannot
    take three steps in one pass (a generic block has no third stage) -- the engine rebuilds its tables for two steps + one
    (Engine::init_walls calls itself), keeps the frame's bricks, and gives the oracle's bits."""
    n = (48, 100, 280)
    src = [n[0] // 2, n[1] // 2, n[2] // 2]
    sim = synth.shoebox(*n, Nt=31, Nm=2, Mb=[11, 3], src=src, rcv=[[src[0] + 2, src[1] - 1, src[2] + 3], [4, 47, 142], [24, 93, 144]], wall=3, rigid_every=13)
    ref = sim_data.SimData.from_sim(sim, "single")
    ref.scale_input()
    e = oracle.Engine(ref)
    for k in range(ref.Nt):
        e.step(k)
    ref_u1 = e.grid(1).copy()
    e.close()
    assert np.abs(ref.u_out).max() > 0
    out, g, tm = run(sim, 40)
    assert tm["tb_steps_per_pass"] == 3 and tm["wall_blocks"][1] > 0 and tm["wall_bricks"] > 0 and tm["wall_three_steps"] == 0, tm
    assert np.array_equal(out, ref.u_out)
    assert np.array_equal(g[1][1:-1, 1:-1, 1:-1], ref_u1[1:-1, 1:-1, 1:-1])


@pytest.mark.parametrize("off,bricks", [(2, True), (1, False)], ids=["two_cells_inside", "one_cell_inside"])
def test_the_frames_bricks_need_the_source_two_cells_inside_the_box(off, bricks):
    """A brick recomputes THREE steps of its halo from u^{n-1}, u^n (pf_brick.h), so a source -- added between the steps -- must stay two
    cells inside the box; one cell inside, the frame goes back to the generic blocks of k_wall2 (two steps + one).  Either way the
    oracle's receivers and field."""
    n = (48, 100, 280)
    sim0 = triple_scene(Nt=40, n=n)
    sd0 = sim_data.SimData.from_sim(sim0, "single", build_mask=False)
    sd0.scale_input()
    eng = engine.HipEngine(sd0, air_variant=40, timing=True)
    tm0 = eng.timing()
    eng.close()
    assert tm0["wall_bricks"] > 0
    # the triples' box starts three cells beyond the deepest wall layer (wall = 3: layers at index 2 and 3, box from 6, Engine::init_tb2_impl);
    # a source `off` cells inside it on x
    x0 = 3 + 3
    src = [x0 + off, n[1] // 2, n[2] // 2]
    sim = synth.shoebox(*n, Nt=40, Nm=2, Mb=[11, 3], src=src, rcv=[[x0 + off + 1, n[1] // 2 + 2, n[2] // 2 - 1], [5, 30, 141], [20, 50, 139]], wall=3)
    ref = sim_data.SimData.from_sim(sim, "single")
    ref.scale_input()
    e = oracle.Engine(ref)
    for k in range(ref.Nt):
        e.step(k)
    ref_u1 = e.grid(1).copy()
    e.close()
    assert np.abs(ref.u_out).max() > 0
    out, g, tm = run(sim, 40)
    assert tm["tb2_launches"] > 0 and sum(tm["wall_blocks"]) > 0, tm
    assert (tm["wall_bricks"] > 0) == bricks and (tm["wall_three_steps"] == 9) == bricks, tm
    assert np.array_equal(out, ref.u_out)
    assert np.array_equal(g[1][1:-1, 1:-1, 1:-1], ref_u1[1:-1, 1:-1, 1:-1])


@pytest.mark.parametrize("prec", ["single", "double"])
def test_three_steps_per_pass_with_geometry_inside_the_box(prec):
    """a block standing in the room: its surface nodes live INSIDE the box of k_tb3 -- their tiles (grown by two cells) take three
    single steps, the nodes three passes of the list kernel (first step beside the wall regions, branch state double-buffered
    with theirs), the neighbouring tiles leave their u^{n+1} behind.  Receivers equal the oracle's, whole fields the single-step
    engine's."""
    n = (48, 100, 280 if prec == "single" else 264)
    sim = triple_scene(Nt=61, n=n, blocks=((14, 18, 30, 40, 60, 130),))
    ref = sim_data.SimData.from_sim(sim, prec)
    ref.scale_input()
    oracle.run_sim(ref)
    assert np.abs(ref.u_out).max() > 0
    _, base_g, _ = run(sim, 25, prec=prec)
    for chunk in (0, 5):
        out, g, tm = run(sim, 40, prec=prec, readout_chunk=chunk)
        assert tm["tb_steps_per_pass"] == 3 and tm["tb2_launches"] > 0 and tm["tb2_dirty_tiles"] >= 2, tm  # (the block's tiles; the source's and the receivers' no longer step singly)
        assert np.array_equal(out, ref.u_out), chunk
        for a, b in zip(g, base_g):
            assert np.array_equal(a[1:-1, 1:-1, 1:-1], b[1:-1, 1:-1, 1:-1]), chunk


def test_placement_search_grows_its_pool_when_no_assignment_is_fast(monkeypatch):
    """Some pools of grids hold no fast assignment for the four streams of a blocked kernel (round 5: 3.40 ms per launch of k_tb3 where
    2.95 is the rule): a search that ends above the known level -- 16 B per cell and launch at 5.5 TB/s -- allocates four more
    candidates and searches on, three times at most.  PFFDTD_PLACE_FORCE_GROW walks that path on any box: more candidates timed, the same
    bits (receivers against the oracle, triples still chosen)."""
    sim = triple_scene(Nt=31)
    ref = sim_data.SimData.from_sim(sim, "single")
    ref.scale_input()
    oracle.run_sim(ref)
    out0, _, tm0 = run(sim, 40)
    monkeypatch.setenv("PFFDTD_PLACE_FORCE_GROW", "1")
    out1, _, tm1 = run(sim, 40)
    assert tm0["tb_steps_per_pass"] == 3 and tm1["tb_steps_per_pass"] == 3
    assert tm1["place_candidates"] > tm0["place_candidates"] > 0, (tm0["place_candidates"], tm1["place_candidates"])
    assert np.array_equal(out0, ref.u_out) and np.array_equal(out1, ref.u_out)
