#!/usr/bin/env python3
"""bench.py -- headline benchmark of the FDTD time-step hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[3]): synthetic shoebox 1024^3, 7-point Cartesian, fp32,
walls 3 cells in, inside wall layer frequency-dependent (Mb=11 branches), outside layer rigid, 1 source,
2 receivers x 8 nodes; state grids pre-filled with seeded U(-1,1)*1e-3 (no all-zero field: see DVFS note in
DESIGN.md).  One "step" = one whole time step (ghost flips, air stencil, ABC, rigid + FD boundary nodes,
source/receiver I/O, slab exchange).  The grid is fixed at 1024^3 for every N (strong scaling: BASELINE.json
names 1024^3 at 1/2/4/8 GPUs); N>1 = Z-slab chain, one rank per GPU, RCCL plane exchange.

Prints ONE JSON line on rank 0: metric Gvoxel-updates/s = Nx*Ny*Nz*K / t / 1e9 (the reference's own formula,
cpu_engine.h:357 / gpu_engine.h:1253), plus `roofline` (air kernel, HIP-event timed, algorithmic bytes =
12.125 B/voxel fp32) and `cpu_baseline` (the CPU oracle on this box's host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def build_scene(n, steps_total, prec, fcc, lossy, mb, nx=0):
    from pffdtd_amd import sim_data, synth
    if fcc:
        # folded FCC with a stored grid of n x n x n: unfolded Ny = 2(n-1)
        sim = synth.shoebox(n, 2 * (n - 1), n, Nt=steps_total, fcc=True, Nm=1, Mb=mb, lossy=lossy)
        synth.fold_fcc(sim)
        synth.sort_sim(sim)
    else:
        sim = synth.shoebox(nx or n, n, n, Nt=steps_total, Nm=1, Mb=mb, lossy=lossy)
    sd = sim_data.SimData.from_sim(sim, prec, build_mask=False)
    sd.scale_input()
    return sd


def usable_cpus():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 quota
        q, p = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(prec, fcc, mb, lossy, budget_s=12.0):
    """The CPU oracle (bit-exact restatement of the reference C CPU engine) on a bounded sample of the same workload:
    a 512^3 room of the same kind (1 GB of state in fp32: far outside the host's caches, like the 1024^3 grid), stepped
    for 500 (fp32) / 250 (fp64) steps on all usable cores: ~12 s of CPU work on the GPU box."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle
    from pffdtd_amd import sim_data, synth
    n = 512
    cores = min(usable_cpus(), 64)

    def run(nt):
        if fcc:
            sim = synth.shoebox(n, 2 * (n - 1), n, Nt=nt, fcc=True, Nm=1, Mb=mb, lossy=lossy)
            synth.fold_fcc(sim)
            synth.sort_sim(sim)
        else:
            sim = synth.shoebox(n, n, n, Nt=nt, Nm=1, Mb=mb, lossy=lossy)
        sd = sim_data.SimData.from_sim(sim, prec)
        sd.scale_input()
        el, t_air, t_bn = oracle.run_sim(sd, threads=cores)
        return sd.Npts, el, t_air

    t0 = time.time()
    nt = 500 if prec == "single" else 250  # 67 / 34 G voxel updates: ~12 s on the GPU box's 16 usable cores (5.8 Gvox/s fp32)
    npts, el, t_air = run(nt)
    wall = time.time() - t0
    if wall > 4 * budget_s:
        print(f"[bench] cpu baseline took {wall:.1f}s", file=sys.stderr)
    return {"value": round(npts * nt / el / 1e9, 4), "unit": "Gvoxel-updates/s", "cores": cores, "kind": "port",
            "sample": f"{n}^3 {'13-pt folded FCC' if fcc else '7-pt Cartesian'} {prec} shoebox, "
                      f"{'Mb=%d lossy walls' % mb if lossy else 'rigid walls'}, {nt} steps = {el:.1f} s, OpenMP CPU oracle "
                      f"(oracle/pf_oracle.c = cpu_engine.h restated, bit-exact vs the compiled reference)",
            "air_fraction": round(t_air / el, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=7, help="the K-step region is timed this many times; value = the MEDIAN region")
    ap.add_argument("--no-rigid-run", action="store_true", help="skip the second, rigid-wall run (SURVEY 8d asks for both)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--nx", type=int, default=0, help="override the number of planes along x (experiments only)")
    ap.add_argument("--precision", default="single", choices=["single", "double"])
    ap.add_argument("--fcc", action="store_true")
    ap.add_argument("--rigid", action="store_true", help="rigid walls only (no FD boundary nodes)")
    ap.add_argument("--mb", type=int, default=11)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--numerics", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--debug", type=lambda v: int(v, 0), default=0, help="pf_opts.debug (tuning switches)")
    ap.add_argument("--split-phase", action="store_true", help="N=1: drive the split-phase step like N>1 does")
    ap.add_argument("--emulate-slab", default="", help="debug: 'r/N' = run only slab r of an N-way split on this GPU, the exchange\n                    replaced by device copies of the same planes (per-rank cost model; physics is wrong)")
    ap.add_argument("--emulate-transport", default="copy", choices=["copy", "rccl"],
                    help="with --emulate-slab: 'rccl' sends the planes to this same rank through RCCL (real launch cost)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")

    # RCCL's intra-node transport shares device memory between the ranks: this pool's host driver only supports dmabuf
    # IPC (without the switch hipIpcGetMemHandle fails); normally exported already, set here as well so that a bare
    # environment cannot break the multi-GPU run
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    from pffdtd_amd import engine
    if not torch.cuda.is_available() or engine.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")
    # debug only: PFFDTD_BENCH_BACKEND=gloo runs every rank on GPU 0 with host-staged planes (control-flow test of the
    # N>1 path on a 1-GPU box); the real multi-GPU run uses RCCL with one GPU per rank
    backend = os.environ.get("PFFDTD_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    group = None
    if args.emulate_slab and args.emulate_transport == "rccl":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29731")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    K, W, R = args.steps, args.warmup, max(args.repeats, 1)
    n = args.size
    lossy = not args.rigid
    sd = build_scene(n, R * K + W, args.precision, args.fcc, lossy, args.mb, args.nx)
    real_bytes = 4 if args.precision == "single" else 8
    ekw = dict(numerics=args.numerics, air_variant=args.variant, air_chunk=args.chunk, timing=True, debug=args.debug)

    from pffdtd_amd import dist as pdist
    emu = None
    if args.emulate_slab:
        er, eN = (int(v) for v in args.emulate_slab.split("/"))
        emu = (er, eN)
        runner, loc, info = pdist.make_hip_runner(sd, er, eN, local_rank, None, **ekw)

        def _fake_exchange(st=runner.st, inf=info):
            s_lo, s_hi, r_lo, r_hi = st.halo_tensors()
            with st.comm_context():
                if not inf.first:
                    r_lo.copy_(s_lo, non_blocking=True)
                if not inf.last:
                    r_hi.copy_(s_hi, non_blocking=True)
        runner.exchange = _fake_exchange
        if args.emulate_transport == "rccl":
            import torch.distributed as dist

            def _self_exchange(st=runner.st, inf=info):
                s_lo, s_hi, r_lo, r_hi = st.halo_tensors()
                ops = []
                if not inf.first:
                    ops += [dist.P2POp(dist.isend, s_lo, 0), dist.P2POp(dist.irecv, r_lo, 0)]
                if not inf.last:
                    ops += [dist.P2POp(dist.isend, s_hi, 0), dist.P2POp(dist.irecv, r_hi, 0)]
                with st.comm_context():
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
            runner.exchange = _self_exchange
    else:
        runner, loc, info = pdist.make_hip_runner(sd, rank, world, local_rank, group, **ekw)
    eng = runner.st.eng
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 + rank)
    for g in runner.st.grids:  # device-side fill of the torch-owned state grids (pad/ghost cells are never read back)
        g.copy_((torch.rand(g.shape, generator=gen, device=g.device, dtype=torch.float32) * 2.0 - 1.0) * 1e-3)
    torch.cuda.synchronize()
    if world == 1 and not args.split_phase and emu is None:
        run = lambda n0, k: eng.run(n0, k)  # noqa: E731  (whole loop inside the C library, one stream)
        parallelism = "1 GPU"
    else:
        run = lambda n0, k: runner.run(n0, k)  # noqa: E731
        parallelism = f"z-slab x{world}, 1 rank/GPU, RCCL p2p plane exchange overlapped with interior planes"
    sync = eng.sync
    timing = eng.timing
    interior_planes = loc.Nx - 2

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    runner.verify_steps = min(W, 4) if world > 1 else 0  # N>1: checksum the first exchanges against the senders' planes
    run(0, W)
    sync()
    timing(reset=True)

    def timed_region(n0):
        """K steps bracketed by barrier + synchronize on both sides; max over ranks"""
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n0, K)
        sync()
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([el], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    regions = [timed_region(W + r * K) for r in range(R)]
    el = sorted(regions)[len(regions) // 2]  # the median region is the one reported
    tm = timing()

    # sanity: the field must still be finite
    if not bool(torch.isfinite(runner.st.grids[0][loc.Nx // 2]).all()):
        raise SystemExit("bench: non-finite field")

    if rank == 0:
        gvox = sd.Npts * K / el / 1e9
        if emu is not None:
            print(f"[emulated slab {emu[0]}/{emu[1]}: {loc.Nx} planes, exchange = {args.emulate_transport}] "
                  f"{el / K * 1e3:.4f} ms/step -> {emu[1]} such ranks would give "
                  f"{sd.Npts * K / el / 1e9:.1f} Gvox/s if the exchange hides completely", file=sys.stderr)
        bpv = 3 * real_bytes + 0.125  # u1 read, u0 read + write, one mask bit
        # interior update of one step on this rank: the voxels it updates, and the time of all its interior launches
        upd = interior_planes * (sd.Ny - 2) * (sd.Nz - 2)
        air_ms_per_step = tm["air_ms_total"] / max(tm["steps"] if tm["steps"] else K, 1)
        if tm.get("tb2_launches", 0) > 0:
            # temporal blocking: the dominant kernel advances `tb2_cells` cells by TWO steps per launch; its algorithmic
            # bytes are therefore 2 x 12.125 B per cell and launch (SURVEY 8d's per-update figure x the updates it performs)
            kernel, kernel_ms = ("k_tb2_fcc" if args.fcc else "k_tb2_reg"), tm["tb2_ms_total"] / tm["tb2_launches"]
            units = 2 * tm["tb2_cells"]
        else:
            kernel = "k_air_fcc" if args.fcc else ("k_air_cart" if tm.get("air_path") == 1 else "k_air_cart_lean")
            kernel_ms, units = air_ms_per_step, upd
        achieved = units * bpv / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None
        res = {
            "metric": "Gvoxel-updates/s", "value": round(gvox, 3), "unit": "Gvoxel-updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(el / K * 1e3, 4),
            "repeats": R, "ms_per_step_min": round(min(regions) / K * 1e3, 4), "ms_per_step_max": round(max(regions) / K * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if real_bytes == 4 else "f64", "data": "synthetic",
            "config": {"workload": f"shoebox {n}^3 {'13-pt folded FCC' if args.fcc else '7-pt Cartesian'} "
                                   f"{'fp32' if real_bytes == 4 else 'fp64'}, "
                                   f"{'Mb=%d freq-dependent walls' % args.mb if lossy else 'rigid walls'} "
                                   "(BASELINE.json configs[3])",
                       "grid": [sd.Nx, sd.Ny, sd.Nz], "Nb": sd.Nb, "Nbl": sd.Nbl, "Nba": sd.Nba,
                       "numerics": "cpu-exact" if args.numerics == 0 else "fma",
                       "parallelism": parallelism, "air_variant": args.variant},
            "achieved_hbm_GBs_whole_step": round(gvox * bpv, 1),
            "whole_step_frac_of_hbm_roofline": round(gvox * bpv / HBM_PEAK_GBS, 4),
            "roofline": {"bound": "hbm", "kernel": kernel,
                         "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": None,
                         "kernel_ms_per_launch": round(kernel_ms, 4), "voxel_updates_per_launch": int(units),
                         "bytes_per_voxel_update": bpv, "air_ms_per_step": round(air_ms_per_step, 4),
                         "interior_voxels_per_step": upd,
                         "autotune_ms_per_step": {k: round(v, 4) for k, v in zip(("lean", "barrier_free", "blocked_pair"), tm.get("tune_ms", [0, 0, 0]))},
                         "grid_placement": {"candidates": tm.get("place_candidates", 0),
                                            "kernel_ms_as_allocated_chosen_slowest": [round(v, 4) for v in tm.get("place_ms", [0, 0, 0])]}},
        }
        # HBM bytes per launch of the dominant kernel: from the committed PMC passes of this same command (rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 x2 read correction applied; tools/collect_n1_profile.sh).  Only
        # quoted when the committed profile is of the very kernel instantiation this run used, advancing the same number of
        # voxel updates per launch -- otherwise null.
        tfile, pfile = ROOT / "profiles" / "r02_bench_n1_hbm_traffic.json", ROOT / "profiles" / "r02_bench_n1.json"
        if world == 1 and tm.get("tb2_launches", 0) > 0 and tfile.exists():
            inst = (f"pf::k_tb2_fcc<{'float' if real_bytes == 4 else 'double'}, 2, 4, {int(tm['tb2_lw'])}>" if args.fcc else
                    f"pf::k_tb2_reg<{'float' if real_bytes == 4 else 'double'}, 3, 4, false, {int(tm['tb2_lw'])}, false>")  # (..., true> = creation-time probes)
            try:
                ks = json.load(open(tfile))["kernels"]
                # (pfile absent = the profile collection itself, tools/collect_n1_profile.sh: the passes just taken are of this build)
                prof = json.load(open(pfile)) if pfile.exists() else None
                hit = [v for k, v in ks.items() if inst in k]
                same = prof is None or (prof["roofline"]["voxel_updates_per_launch"] == int(units) and prof["config"]["grid"] == [sd.Nx, sd.Ny, sd.Nz]
                        and prof["config"]["Nb"] == sd.Nb and prof["dtype"] == res["dtype"])
                if hit and same:
                    res["roofline"]["traffic"] = round(hit[0]["total_bytes"] / 1e9, 3)
                    res["roofline"]["traffic_unit"] = f"GB per launch of {inst} (PMC, profiles/r02_bench_n1_hbm_traffic.json)"
                    res["roofline"]["algorithmic_GB_per_launch"] = round(units * bpv / 1e9, 3)
                    res["roofline"]["measured_traffic_GBs"] = round(hit[0]["total_bytes"] / 1e9 / (kernel_ms * 1e-3), 1)
                    res["roofline"]["note"] = ("a launch advances its cells by TWO steps: achieved = 2 x 12.125 B per cell / launch time (SURVEY 8d's per-update "
                                               "figure x the updates of a launch), which temporal blocking is allowed to beat; measured_traffic_GBs = the "
                                               "HBM bytes the launch really moves / launch time")
                else:
                    res["roofline"]["traffic_note"] = "committed profile is of another kernel instantiation / workload: not quoted"
            except (OSError, KeyError, ValueError):
                pass
        if world > 1:
            res["exchange_verified"] = runner.exchange_verified
            res["exchange"] = {"backend": backend, "ranks": world, "checked_steps": min(W, 4),
                               "what": "bit-pattern checksums of the received ghost planes == the senders' planes, all ranks"}
        if world == 1 and lossy and not args.no_rigid_run and emu is None:
            # a rigid-wall run next to the frequency-dependent one (same grid, no branch ODEs)
            runner.st.close()
            runner.st.grids.clear()
            torch.cuda.empty_cache()
            sd_r = build_scene(n, 3 * K + W, args.precision, args.fcc, False, args.mb, args.nx)
            rr, loc_r, _ = pdist.make_hip_runner(sd_r, 0, 1, local_rank, None, **ekw)
            for g in rr.st.grids:
                g.copy_((torch.rand(g.shape, generator=gen, device=g.device, dtype=torch.float32) * 2.0 - 1.0) * 1e-3)
            torch.cuda.synchronize()
            rr.st.eng.run(0, W)
            rr.st.eng.sync()
            ts = []
            for r in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rr.st.eng.run(W + r * K, K)
                rr.st.eng.sync()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            t_r = sorted(ts)[1]
            tm_r = rr.st.eng.timing()
            res["rigid_walls"] = {"value": round(sd_r.Npts * K / t_r / 1e9, 3), "unit": "Gvoxel-updates/s",
                                  "ms_per_step": round(t_r / K * 1e3, 4), "repeats": 3, "Nb": sd_r.Nb,
                                  "whole_step_frac_of_hbm_roofline": round(sd_r.Npts * K / t_r / 1e9 * bpv / HBM_PEAK_GBS, 4),
                                  "blocked_pairs": tm_r.get("tb2_launches", 0) > 0,
                                  "autotune_ms_per_step": {k: round(v, 4) for k, v in zip(("lean", "barrier_free", "blocked_pair"), tm_r.get("tune_ms", [0, 0, 0]))}}
            rr.st.close()
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args.precision, args.fcc, args.mb, lossy)
        print(json.dumps(res), flush=True)

    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
