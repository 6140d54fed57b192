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
names 1024^3 at 1/2/4/8 GPUs); N>1 = Z-slab chain with a one-plane exchange per step, in one of two arrangements:

  * under torch.distributed.run (WORLD_SIZE == N): one rank per GPU, RCCL p2p through torch.distributed;
  * plain `python bench.py --gpus N` (no launcher): ONE process drives devices 0..N-1 through the C seam's chain object
    (pf_multi_create: one host thread per slab, ghost planes by peer copies or native RCCL) -- the reference's own
    arrangement (gpu_engine.h:680-682).  On a box with fewer than N devices the slabs share devices ("virtual slabs").

Prints ONE JSON line on rank 0: metric Gvoxel-updates/s = Nx*Ny*Nz*K / t / 1e9 (the reference's own formula,
cpu_engine.h:357 / gpu_engine.h:1253), plus `roofline` (dominant kernel, HIP-event timed in a region of its own --
the headline regions run without per-launch events --, algorithmic bytes = 12.125 B/voxel fp32) and `cpu_baseline`
(the CPU oracle on this box's host cores, bounded sample).
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
FAMILY = {0: "lean single steps", 1: "barrier-free single steps", 2: "blocked pairs"}


def build_scene(n, steps_total, prec, fcc, lossy, mb, nx=0, ny=0):
    from pffdtd_amd import sim_data, synth
    if fcc:
        # folded FCC with a stored grid of n x n x n: unfolded Ny = 2(n-1)
        sim = synth.shoebox(n, 2 * (n - 1), n, Nt=steps_total, fcc=True, Nm=1, Mb=mb, lossy=lossy)
        synth.fold_fcc(sim)
        synth.sort_sim(sim)
    else:
        sim = synth.shoebox(nx or n, ny or n, n, Nt=steps_total, Nm=1, Mb=mb, lossy=lossy)
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
    """The reference C CPU engine on a bounded sample of the same workload: a 512^3 room of the same kind (1 GB of state in
    fp32: far outside the host's caches, like the 1024^3 grid), stepped for 500 (fp32) / 250 (fp64) steps on all usable cores:
    ~12 s of CPU work on the GPU box.  kind = "reference": the reference's own binary (c_cuda/fdtd_main.c compiled by
    oracle/Makefile into oracle/_ref, which travels to the GPU box as a built file) run in a folder written by
    synth.write_folder, its own `Combined (total)` line parsed (cpu_engine.h:355-357); kind = "port" where that binary or
    libhdf5 is missing: the CPU oracle (oracle/pf_oracle.c, the same loop restated, bit-pinned to that binary)."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle
    from pffdtd_amd import sim_data, synth
    n = 512
    cores = min(usable_cpus(), 64)
    nt = 500 if prec == "single" else 250  # 67 / 34 G voxel updates: ~12 s on the GPU box's 16 usable cores (5.8 Gvox/s fp32)
    if fcc:
        sim = synth.shoebox(n, 2 * (n - 1), n, Nt=nt, fcc=True, Nm=1, Mb=mb, lossy=lossy)
        synth.fold_fcc(sim)
        synth.sort_sim(sim)
    else:
        sim = synth.shoebox(n, n, n, Nt=nt, Nm=1, Mb=mb, lossy=lossy)
    what = (f"{n}^3 {'13-pt folded FCC' if fcc else '7-pt Cartesian'} {prec} shoebox, "
            f"{'Mb=%d lossy walls' % mb if lossy else 'rigid walls'}, {nt} steps")
    t0 = time.time()
    ref_note = "oracle/_ref not on this box"
    if oracle.ref_binary(prec) is not None:
        import re
        import shutil
        import subprocess
        import tempfile
        td = tempfile.mkdtemp(prefix="pf_cpu_baseline_", dir="/tmp")
        try:
            synth.write_folder(sim, td)
            with open(Path(td) / "stdout.log", "w") as log:  # (the reference redraws a progress block every step: to a file)
                r = subprocess.run([str(oracle.ref_binary(prec))], cwd=td, stdout=log, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL,
                                   env={**os.environ, "OMP_NUM_THREADS": str(cores), "TERM": "dumb"}, timeout=20 * budget_s)
            tail = open(Path(td) / "stdout.log", errors="replace").read()
            mt = re.search(r"Combined \(total\): ([0-9.eE+-]+)s, ([0-9.eE+-]+) Mvox/s", tail)
            ma = re.search(r"Air update: ([0-9.eE+-]+)s", tail)
            if r.returncode == 0 and mt:
                el = float(mt.group(1))
                return {"value": round(float(mt.group(2)) / 1e3, 4), "unit": "Gvoxel-updates/s", "cores": cores, "kind": "reference",
                        "sample": f"{what} = {el:.1f} s: the reference's own c_cuda/fdtd_main.c (CPU engine, OpenMP, compiled unmodified by "
                                  f"oracle/Makefile -> oracle/_ref/{oracle.ref_binary(prec).name}) on a folder written by synth.write_folder, "
                                  f"OMP_NUM_THREADS={cores}; value = its own 'Combined (total)' line (cpu_engine.h:357)",
                        "air_fraction": round(float(ma.group(1)) / el, 3) if ma else None}
            ref_note = f"reference binary failed here (rc {r.returncode})"
        except (OSError, subprocess.SubprocessError, RuntimeError) as e:
            ref_note = f"reference binary not usable here ({type(e).__name__})"
        finally:
            shutil.rmtree(td, ignore_errors=True)
    sd = sim_data.SimData.from_sim(sim, prec)
    sd.scale_input()
    el, t_air, t_bn = oracle.run_sim(sd, threads=cores)
    wall = time.time() - t0
    if wall > 4 * budget_s:
        print(f"[bench] cpu baseline took {wall:.1f}s", file=sys.stderr)
    return {"value": round(sd.Npts * nt / el / 1e9, 4), "unit": "Gvoxel-updates/s", "cores": cores, "kind": "port",
            "sample": f"{what} = {el:.1f} s, OpenMP CPU oracle (oracle/pf_oracle.c = cpu_engine.h restated, bit-exact vs the compiled "
                      f"reference; {ref_note})",
            "air_fraction": round(t_air / el, 3)}


class _DevMem:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def fill_engine_grids(torch, eng_view, nplanes, plane_elems, real_bytes, device, seed):
    """Seeded U(-1,1)*1e-3 into both state grids of an engine, written on the device through torch views of its allocations."""
    ts, dt = ("<f4", torch.float32) if real_bytes == 4 else ("<f8", torch.float64)
    gen = torch.Generator(device=f"cuda:{device}")
    gen.manual_seed(seed)
    with torch.cuda.device(device):
        for p in eng_view.state_grids():
            g = torch.as_tensor(_DevMem(p, (nplanes, plane_elems), ts), device=f"cuda:{device}")
            assert g.data_ptr() == p
            g.copy_(((torch.rand(g.shape, generator=gen, device=g.device, dtype=torch.float32) * 2.0 - 1.0) * 1e-3).to(dt))
            del g
        torch.cuda.synchronize()


def roofline_block(args, sd, tm, K, real_bytes, interior_planes, interior_per_plane=None):
    """`roofline` of the bench line from an engine's timing record (collected in a region with per-launch events on).
    interior_per_plane: interior cells of one stored plane (default: the file's (Ny-2)(Nz-2); a chain cut along file z
    stores planes of Ny x Nx)."""
    bpv = 3 * real_bytes + 0.125  # u1 read, u0 read + write, one mask bit
    T = "float" if real_bytes == 4 else "double"
    upd = interior_planes * (interior_per_plane if interior_per_plane is not None else (sd.Ny - 2) * (sd.Nz - 2))
    air_ms_per_step = tm["air_ms_total"] / max(tm["steps"] if tm["steps"] else K, 1)
    if tm.get("tb2_launches", 0) > 0:
        # temporal blocking: the dominant kernel advances `tb2_cells` cells by TWO steps per launch; its algorithmic
        # bytes are therefore 2 x 12.125 B per cell and launch (SURVEY 8d's per-update figure x the updates it performs)
        lw = int(tm["tb2_lw"])
        spp = int(tm.get("tb_steps_per_pass") or 2)  # steps a launch advances its cells by: 2 (pairs) or 3 (k_tb3)
        if spp == 3:
            sgt = "true" if getattr(args, "numerics", 0) == 2 else "false"  # (<..., PROBE = true> = creation-time probes)
            kernel, inst = "k_tb3", f"pf::k_tb3<{T}, 3, 8, {sgt}, false, 3>"
        elif args.fcc:
            sgt = "true" if getattr(args, "numerics", 0) == 2 else "false"
            old_kernel = bool(getattr(args, "debug", 0) & 0x40000) and sgt == "false"
            kernel = ("k_tb2_fcc_x" if old_kernel else "k_tb2_fcc_w") if lw == 64 else "k_tb2_fcc"
            inst = ((f"pf::k_tb2_fcc_x<{T}, 2, 8>" if old_kernel else f"pf::k_tb2_fcc_w<{T}, 2, 8, {sgt}, false>") if lw == 64
                    else f"pf::k_tb2_fcc<{T}, 2, 4, {lw}, {sgt}, false>")
        else:
            sgt = "true" if getattr(args, "numerics", 0) == 2 else "false"  # (<..., PROBE = true, ...> = creation-time probes)
            kernel, inst = "k_tb2_reg", f"pf::k_tb2_reg<{T}, 3, 4, false, {lw}, false, {sgt}>"
        kernel_ms = tm["tb2_ms_total"] / tm["tb2_launches"]
        units = spp * tm["tb2_cells"]
    else:
        kernel = "k_air_fcc" if args.fcc else ("k_air_cart" if tm.get("air_path") == 1 else "k_air_cart_lean")
        inst = f"pf::{kernel}<{T},"
        kernel_ms, units = air_ms_per_step, upd
    achieved = units * bpv / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None
    rl = {"bound": "hbm", "kernel": kernel, "kernel_instantiation": inst,
          "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": None,
          "kernel_ms_per_launch": round(kernel_ms, 4), "voxel_updates_per_launch": int(units),
          "bytes_per_voxel_update": bpv, "air_ms_per_step": round(air_ms_per_step, 4),
          "interior_voxels_per_step": upd,
          "timed_how": "HIP events around every launch of the kernel, in a K-step region of its own after the headline regions (which run without per-launch events)",
          "autotune_ms_per_step": {k: round(v, 4) for k, v in zip(("lean", "barrier_free", "blocked_pair"), tm.get("tune_ms", [0, 0, 0]))},
          "grid_placement": {"candidates": tm.get("place_candidates", 0),
                             "kernel_ms_as_allocated_chosen_slowest": [round(v, 4) for v in tm.get("place_ms", [0, 0, 0])]}}
    if tm.get("tb2_launches", 0) > 0 and kernel_ms > 0:
        # what a two- (three-) steps-per-pass kernel MUST move: u^{n-1}, u^n read once, two time levels written once = 4 values per
        # cell and launch -- a fraction of a hardware limit (<= 1 by construction), unlike `frac` above
        rl["steps_per_launch"] = int(tm.get("tb_steps_per_pass") or 2)
        comp = tm["tb2_cells"] * 4 * real_bytes
        rl["blocked_compulsory_GB_per_launch"] = round(comp / 1e9, 3)
        rl["frac_of_blocked_compulsory"] = round(comp / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        wb = tm.get("wall_blocks", [0, 0])
        nbr, w3 = int(tm.get("wall_bricks", 0)), int(tm.get("wall_three_steps", 0)) & 0xf  # (bits 4, 5: an end slab's x wall)
        if not sum(wb):
            rl["shell"] = "single steps"
        else:
            frame = (f"{nbr} bricks stepped in LDS (k_brick)" if nbr else f"{wb[1]} generic blocks of k_wall2")
            if rl["steps_per_launch"] == 3:
                how = ("all three steps in ONE pass before the box kernel (k_wall2 NS = 3: x / y regions and column strips)" if w3 == 9 else
                       "x / y regions three steps in one pass (k_wall2 NS = 3), column strips two steps + one" if w3 == 1 else
                       "two steps as wall regions + " + ("one single step by the list kernels" if ((getattr(args, "debug", 0) & 0x80000) or getattr(args, "gpus", 1) > 1 or getattr(args, "emulate_slab", "")) else "the regions' one-step form (k_wall2 NS = 1)"))
            else:
                how = "both steps of a pair in one pass (k_wall2 NS = 2)"
            rl["shell"] = f"wall regions (k_wall2): {wb[0]} blocks of alike pencils, {how}; frame (edges, corners): {frame}"
    return rl, bpv, kernel_ms, units


def measure_traffic_live(args, inst):
    """HBM bytes per launch of the dominant kernel, measured NOW on this box: two child passes of this same command under
    rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in separate runs, kernel trace only, as guides/MI355X_MICROARCH.md prescribes;
    gfx950: FETCH_SIZE x 2 for wide coalesced reads), a few steps each, no creation-time measurement (debug 0x8000: under counter
    collection it can reject the path the timed run uses).  None when rocprofv3 is not on the box or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if Path("/opt/rocm/bin/rocprofv3").exists() else None)
    if not rp:
        return {"failed": "rocprofv3 not on this box"}
    try:
        import torch
        arch = torch.cuda.get_device_properties(0).gcnArchName.split(":")[0]
    except Exception:  # noqa: BLE001
        arch = ""
    rd_scale = 2 if arch == "gfx950" else 1  # gfx950 tallies 128-B read requests as 64 B (guides/MI355X_MICROARCH.md)
    child = [sys.executable, str(ROOT / "bench.py"), "--steps", "6", "--warmup", "2", "--repeats", "1", "--no-rigid-run", "--no-cpu-baseline",
             "--no-selfcheck", "--no-pmc", "--size", str(args.size), "--precision", args.precision, "--mb", str(args.mb),
             "--variant", str(args.variant), "--chunk", str(args.chunk), "--numerics", str(args.numerics), "--debug", hex(args.debug | 0x8000)]
    child += ["--split-phase"] if args.split_phase else []
    child += (["--fcc"] if args.fcc else []) + (["--rigid"] if args.rigid else []) + (["--nx", str(args.nx)] if args.nx else []) + (["--ny", str(args.ny)] if args.ny else [])
    out = {}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = Path(td) / counter
            try:
                r = subprocess.run([rp, "--pmc", counter, "--kernel-trace", "-d", str(d), "-o", "p", "--output-format", "csv", "--"] + child,
                                   cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, capture_output=True, text=True, timeout=60)  # (a pass takes ~5 s; a hung profiler must not hold the bench)
            except (OSError, subprocess.TimeoutExpired) as e:
                return {"failed": f"{counter} pass: {type(e).__name__}"}
            vals = []
            for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if inst in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or not vals:
                return {"failed": f"{counter} pass: rc {r.returncode}, {len(vals)} rows of {inst}"}
            out[counter] = sorted(vals)[len(vals) // 2]
    rd, wr = out["FETCH_SIZE"] * 1024 * rd_scale, out["WRITE_SIZE"] * 1024  # KiB -> bytes
    return {"read_bytes": rd, "write_bytes": wr, "total_bytes": rd + wr, "seconds": round(time.perf_counter() - t0, 1), "arch": arch, "read_scale": rd_scale}


def bounded_headline(rl):
    """A kernel that advances its cells by more than one step per launch beats SURVEY 8d's per-update byte figure by design, so
    that figure over the peak is no fraction of a hardware limit (round-5 verdict).  For such kernels `achieved` / `frac` are the
    HBM bytes the launch really moved (PMC) over its duration; the 8d figure stays beside them as *_algorithmic_8d."""
    if rl.get("steps_per_launch", 1) > 1 and "frac_algorithmic_8d" not in rl:
        rl["achieved_algorithmic_8d"], rl["frac_algorithmic_8d"] = rl["achieved"], rl["frac"]
        if rl.get("measured_traffic_frac") is not None:
            rl["achieved"], rl["frac"] = rl["measured_traffic_GBs"], rl["measured_traffic_frac"]
            rl["frac_is"] = "PMC bytes per launch / launch time / peak (steps_per_launch > 1: the 8d figure is under frac_algorithmic_8d)"
        else:  # no counter pass of this instantiation (--no-pmc, rocprofv3 unusable): the bytes such a launch MUST move, a lower bound of the above
            rl["frac"] = rl["frac_of_blocked_compulsory"]
            rl["achieved"] = round(rl["frac"] * HBM_PEAK_GBS, 1)
            rl["frac_is"] = ("compulsory bytes of a multi-step launch (4 grid passes) / launch time / peak -- no PMC traffic of this instantiation in "
                             "this run; the 8d figure is under frac_algorithmic_8d")


def add_traffic(rl, res, sd, kernel_ms, units, bpv, live=None):
    """HBM bytes per launch of the dominant kernel: from the committed PMC passes of this same command (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 x2 read correction applied; tools/collect_n1_profile.sh) -- counters
    cannot be read from inside the process.  Only quoted when the committed profile is of the very kernel instantiation this
    run launched, advancing the same number of voxel updates per launch on the same scene; otherwise null."""
    inst = rl["kernel_instantiation"]
    note = (f"a launch advances its cells by {rl.get('steps_per_launch', 2)} steps: achieved = steps x 12.125 B per cell / launch time (SURVEY 8d's per-update "
            "figure x the updates of a launch) is kept as achieved_algorithmic_8d / frac_algorithmic_8d (temporal blocking is allowed to beat it: it "
            "may exceed 1); achieved / frac = measured_traffic_GBs / measured_traffic_frac = the HBM bytes the launch really moves (PMC) / "
            "launch time, over the 8 TB/s peak")
    if live and live.get("failed"):
        rl["traffic_live_attempt"] = "failed: " + live["failed"]  # (the committed passes below stand in, and say so)
        live = None
    if live:
        rl["traffic"] = round(live["total_bytes"] / 1e9, 3)
        rl["traffic_source"] = (f"measured in this run: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE; {live['arch']}: reads x {live['read_scale']}) of this "
                                f"same workload on this box, {live['seconds']} s")
        rl["traffic_unit"] = f"GB per launch of {inst} (PMC: {live['read_bytes'] / 1e9:.3f} read + {live['write_bytes'] / 1e9:.3f} written)"
        rl["algorithmic_GB_per_launch"] = round(units * bpv / 1e9, 3)
        rl["measured_traffic_GBs"] = round(live["total_bytes"] / 1e9 / (kernel_ms * 1e-3), 1)
        rl["measured_traffic_frac"] = round(rl["measured_traffic_GBs"] / HBM_PEAK_GBS, 4)  # the fraction of a hardware limit
        if "k_tb" in inst:
            rl["note"] = note
        bounded_headline(rl)
        return
    for tag in ("r06", "r05", "r04", "r03", "r02"):
        tfile, pfile = ROOT / "profiles" / f"{tag}_bench_n1_hbm_traffic.json", ROOT / "profiles" / f"{tag}_bench_n1.json"
        if tfile.exists():
            break
    else:
        bounded_headline(rl)
        return
    try:
        ks = json.load(open(tfile))["kernels"]
        # (pfile absent = the profile collection itself: the passes just taken are of this build)
        prof = json.load(open(pfile)) if pfile.exists() else None
        hit = [v for k, v in ks.items() if inst in k]
        same = prof is None or (prof["roofline"]["voxel_updates_per_launch"] == int(units) and prof["config"]["grid"] == [sd.Nx, sd.Ny, sd.Nz]
                                and prof["config"]["Nb"] == sd.Nb and prof["dtype"] == res["dtype"])
        if hit and same:
            rl["traffic"] = round(hit[0]["total_bytes"] / 1e9, 3)
            rl["traffic_source"] = f"committed: profiles/{tfile.name} (PMC passes of this same command on another run; rocprofv3 not usable in this one)"
            rl["traffic_unit"] = f"GB per launch of {inst} (PMC, profiles/{tfile.name})"
            rl["algorithmic_GB_per_launch"] = round(units * bpv / 1e9, 3)
            rl["measured_traffic_GBs"] = round(hit[0]["total_bytes"] / 1e9 / (kernel_ms * 1e-3), 1)
            rl["measured_traffic_frac"] = round(rl["measured_traffic_GBs"] / HBM_PEAK_GBS, 4)  # the fraction of a hardware limit
            rl["note"] = note
            bounded_headline(rl)
        else:
            rl["traffic_note"] = "committed profile is of another kernel instantiation / workload: not quoted"
    except (OSError, KeyError, ValueError):
        pass
    bounded_headline(rl)


def base_result(args, sd, world, K, W, R, regions, el, real_bytes, lossy, parallelism):
    gvox = sd.Npts * K / el / 1e9
    bpv = 3 * real_bytes + 0.125
    n = args.size
    # which BASELINE.json configuration is this?  (configs[3]: shoebox 1024^3 7-pt fp32; configs[4]: 1536^3 13-pt folded FCC fp64)
    if not args.fcc and real_bytes == 4 and n == 1024 and not (args.nx or args.ny):
        which = "BASELINE.json configs[3]" + ("" if lossy and args.mb == 11 else ", wall variant")
    elif args.fcc and real_bytes == 8 and n == 1536:
        which = "BASELINE.json configs[4] on this many GPUs" + ("" if lossy and args.mb == 11 else ", wall variant")
    else:
        which = "an experiment, no BASELINE.json configuration"
    return {
        "metric": "Gvoxel-updates/s", "value": round(gvox, 3), "unit": "Gvoxel-updates/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(el / K * 1e3, 4),
        "repeats": R, "ms_per_step_min": round(min(regions) / K * 1e3, 4), "ms_per_step_max": round(max(regions) / K * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if real_bytes == 4 else "f64", "data": "synthetic",
        "config": {"workload": f"shoebox {n}^3 {'13-pt folded FCC' if args.fcc else '7-pt Cartesian'} "
                               f"{'fp32' if real_bytes == 4 else 'fp64'}, "
                               f"{'Mb=%d freq-dependent walls' % args.mb if lossy else 'rigid walls'} "
                               f"({which})",
                   "grid": [sd.Nx, sd.Ny, sd.Nz], "Nb": sd.Nb, "Nbl": sd.Nbl, "Nba": sd.Nba,
                   "numerics": {0: "cpu-exact", 1: "fma", 2: "safeguarded"}.get(args.numerics, str(args.numerics)),
                   "parallelism": parallelism, "air_variant": args.variant},
        "achieved_hbm_GBs_whole_step": round(gvox * bpv, 1),
        "whole_step_frac_of_hbm_roofline": round(gvox * bpv / HBM_PEAK_GBS, 4),
    }


# ------------------------------------------------------------------------------------------------------------------
# N > 1 from ONE plain process: the C seam's chain object (pf_multi_*), one host thread per slab
# ------------------------------------------------------------------------------------------------------------------
def run_chain(args, only=None):
    """only = (r, N): the cost model of rank r of an N-rank chain on ONE device -- the chain is cut as usual, slab r alone is
    instantiated (pf_opts.only_slab) and exchanges its own edge planes with itself through the chosen transport."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    from pffdtd_amd import engine
    ndev = engine.device_count()
    if not torch.cuda.is_available() or ndev == 0:
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")
    N, K, W, R = args.gpus, args.steps, args.warmup, max(args.repeats, 1)
    if only:
        N = only[1]
    devices = [0] * N if only else (list(range(N)) if ndev >= N else [i % ndev for i in range(N)])
    lossy = not args.rigid
    real_bytes = 4 if args.precision == "single" else 8
    sd = build_scene(args.size, (R + 1) * K + W, args.precision, args.fcc, lossy, args.mb, args.nx, args.ny)
    transport = {"auto": engine.PF_TRANSPORT_AUTO, "peer": engine.PF_TRANSPORT_PEER, "rccl": engine.PF_TRANSPORT_RCCL,
                 "host": engine.PF_TRANSPORT_HOST}[args.transport]
    if only:
        transport = engine.PF_TRANSPORT_RCCL if args.emulate_transport == "rccl" else engine.PF_TRANSPORT_PEER
    m = engine.HipMulti(sd, devices, numerics=args.numerics, air_variant=args.variant, air_chunk=args.chunk, debug=args.debug,
                        transport=transport, verify_exchange=0 if only else min(max(W, 2), 6), only_slab=(only[0] + 1) if only else 0, wall_scale=args.wall_scale,
                        multi_flags=0 if args.wall_scale > 0 else engine.PF_MULTI_MEASURE_WEIGHTS)  # (first contact: the first exchanges of every chain -- up to six: two whole triples' worth of split-phase steps -- are checksummed, inside the warm-up)
    live = [only[0]] if only else list(range(N))
    slabs = [m.slab(g) for g in live]
    for g, sl in zip(live, slabs):
        (nloc, ny, _), pitch, _ = sl["engine"].layout()
        fill_engine_grids(torch, sl["engine"], nloc, ny * pitch, real_bytes, sl["device"], 1234 + g)

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    m.run(0, W)

    def timed_region(n0):
        sync_all()
        t0 = time.perf_counter()
        m.run(n0, K)  # returns when every slab's streams have drained
        sync_all()
        return time.perf_counter() - t0

    regions = [timed_region(W + r * K) for r in range(R)]
    el = sorted(regions)[len(regions) // 2]
    info = m.info()
    if not only and not np.isfinite(sd.u_out[:, :W + R * K]).all():
        raise SystemExit("bench: non-finite receiver samples")
    # kernel durations: one more region with per-launch events on
    for sl in slabs:
        sl["engine"].set_timing(True)
        sl["engine"].timing(reset=True)
    m.run(W + R * K, K)
    tms = [sl["engine"].timing() for sl in slabs]
    virt = ndev < N
    parallelism = (f"z-slab x{N}, ONE process, one host thread per slab (pf_multi_create), ghost planes by {info['transport_name']} "
                   "on the edge stream while the interior planes run"
                   + (f"; VIRTUAL: the {N} slabs share {ndev} device(s) -- control-flow run, not a scaling figure" if virt else ""))
    res = base_result(args, sd, N, K, W, R, regions, el, real_bytes, lossy, parallelism)
    if only:
        parallelism = (f"COST MODEL of rank {only[0]} of a z-slab x{N} chain on one device: slab {only[0]} alone exists (pf_opts.only_slab) and "
                       f"exchanges its own edge planes with itself by {info['transport_name']}; value = what an {N}-rank chain of such ranks would do")
        res = base_result(args, sd, N, K, W, R, regions, el, real_bytes, lossy, parallelism)
        res["emulated_slab"] = {"rank": only[0], "of": N, "planes": [slabs[0]["x0"], slabs[0]["x1"]], "pairs": slabs[0]["paired"], "steps_per_pass": slabs[0]["steps_per_pass"],
                                "ms_per_step": round(el / K * 1e3, 4), "ideal_ms_per_step_note": "single-domain ms/step / N"}
    g0 = max(range(len(slabs)), key=lambda g: slabs[g]["x1"] - slabs[g]["x0"])
    (_, ly, lz), _, _ = slabs[g0]["engine"].layout()  # stored rows x columns of a plane (exchanged axes: Ny x Nx)
    rl, bpv, kernel_ms, units = roofline_block(args, sd, tms[g0], K, real_bytes, slabs[g0]["x1"] - slabs[g0]["x0"], (ly - 2) * (lz - 2))
    rl["slab"] = g0
    bounded_headline(rl)
    res["roofline"] = rl
    res["exchange_verified"] = None if only else info["exchange_verified"]
    res["transport"] = info["transport_name"]  # "peer copies" | "rccl" | "host-staged" (the last resort: neither peer access nor a working RCCL)
    if info.get("transport_note"):
        res["transport_note"] = info["transport_note"]
    res["partition"] = {"wall_scale": round(info["wall_scale"], 3), "measured_at_creation": info["wall_measured"],
                        "planes": [s["x1"] - s["x0"] for s in slabs] if not only else None}
    res["exchange"] = {"backend": info["transport_name"], "ranks": N, "checked_steps": info["exchanges_checked"],
                       "nonzero_planes": info["exchange_nonzero"], "plane_bytes": info["plane_bytes"],
                       "what": "position-weighted bit-pattern checksums of the received ghost planes == the senders' planes, every slab"}
    res["slabs"] = [{"device": sl["device"], "planes": [sl["x0"], sl["x1"]], "pairs": sl["paired"], "steps_per_pass": sl["steps_per_pass"],
                     "air_ms_per_step": round(t["air_ms_total"] / K, 4), "wall_region_blocks": sum(t.get("wall_blocks", [0, 0])),
                     "wall_bricks": int(t.get("wall_bricks", 0)), "wall_three_steps": int(t.get("wall_three_steps", 0))}
                    for sl, t in zip(slabs, tms)]
    res["virtual_slabs"] = virt
    m.close()
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=7, help="the K-step region is timed this many times; value = the MEDIAN region")
    ap.add_argument("--no-rigid-run", action="store_true", help="skip the second, rigid-wall run (SURVEY 8d asks for both)")
    ap.add_argument("--no-selfcheck", action="store_true", help="skip the family-agreement check of the timed run")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--nx", type=int, default=0, help="override the number of planes along x (experiments only)")
    ap.add_argument("--ny", type=int, default=0, help="override the number of rows along y (experiments only)")
    ap.add_argument("--precision", default="single", choices=["single", "double"])
    ap.add_argument("--fcc", action="store_true")
    ap.add_argument("--rigid", action="store_true", help="rigid walls only (no FD boundary nodes)")
    ap.add_argument("--mb", type=int, default=11)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--numerics", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 --pmc child passes (then: the committed profile's figure)")
    ap.add_argument("--debug", type=lambda v: int(v, 0), default=0, help="internal PF_DBG_* switches (csrc/pf_debug.h), through the library's internal hook")
    ap.add_argument("--transport", default="auto", choices=["auto", "peer", "rccl", "host"],
                    help="N>1 from one process: how ghost planes travel (pf_opts.transport)")
    ap.add_argument("--wall-scale", type=float, default=1.0, help="N>1 from one process: factor on the wall planes' weights of the cut "
                    "(pf_opts.wall_scale); 1 = the compiled-in weights, 0 = measured on the scene when the chain is created (PF_MULTI_MEASURE_WEIGHTS)")
    ap.add_argument("--split-phase", action="store_true", help="N=1: drive the split-phase step like N>1 does")
    ap.add_argument("--emulate-slab", default="", help="debug: 'r/N' = run only slab r of an N-way split on this GPU, the exchange\n                    replaced by device copies of the same planes (per-rank cost model; physics is wrong)")
    ap.add_argument("--emulate-transport", default="copy", choices=["copy", "rccl", "native"],
                    help="with --emulate-slab: 'rccl' sends the planes to this same rank through RCCL (real launch cost)")
    ap.add_argument("--emulate-via", default="chain", choices=["chain", "torch"],
                    help="with --emulate-slab: through the C chain object (pf_opts.only_slab; default) or the torch.distributed runner")
    args = ap.parse_args()
    if args.emulate_slab and args.emulate_via == "chain":
        return run_chain(args, only=tuple(int(v) for v in args.emulate_slab.split("/")))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            return run_chain(args)  # no launcher: one process drives every device through the C seam
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
    # (native too: the process then holds a torch RCCL process group, as every rank under torch.distributed.run does -- measured: without one
    # the library's own communicator exchanges 0.08 ms per step slower, 0.305 against 0.228 for a rank of 8)
    if args.emulate_slab and args.emulate_transport in ("rccl", "native"):
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
    nt_total = (R + 1) * K + W  # warm-up, R headline regions, one region with per-launch events
    sd = build_scene(n, nt_total, args.precision, args.fcc, lossy, args.mb, args.nx, args.ny)
    real_bytes = 4 if args.precision == "single" else 8
    ekw = dict(numerics=args.numerics, air_variant=args.variant, air_chunk=args.chunk, timing=False, debug=args.debug)

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
        elif args.emulate_transport == "native":  # (--emulate-via torch) the library's own ncclSend / ncclRecv, a 1-rank communicator
            del runner.exchange                   # (the class's method again)
            if not runner.enable_native_rccl(local_rank, peers=(-1 if info.first else 0, -1 if info.last else 0)):
                raise SystemExit("bench: native RCCL unavailable: " + getattr(runner, "native_note", "?"))
    else:
        runner, loc, info = pdist.make_hip_runner(sd, rank, world, local_rank, group, **ekw)
    eng = runner.st.eng

    def fill(st, seed):
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed)
        for g in st.grids:  # device-side fill of the state grids (pad/ghost cells are never read back)
            g.copy_((torch.rand(g.shape, generator=gen, device=g.device, dtype=torch.float32) * 2.0 - 1.0) * 1e-3)
        torch.cuda.synchronize()

    fill(runner.st, 1234 + rank)
    single = world == 1 and not args.split_phase and emu is None
    if single:
        run = lambda n0, k: eng.run(n0, k)  # noqa: E731  (whole loop inside the C library, one stream)
        parallelism = "1 GPU"
    else:
        run = lambda n0, k: runner.run(n0, k)  # noqa: E731
        parallelism = f"z-slab x{world}, 1 rank/GPU, RCCL p2p plane exchange overlapped with interior planes"
    sync = eng.sync
    interior_planes = loc.Nx - 2

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    runner.verify_steps = min(W, 6) if world > 1 else 0  # N>1: checksum the first exchanges against the senders' planes
    run(0, W)
    sync()

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
    # kernel durations: one more K-step region with HIP events around every launch (not part of `value`)
    eng.set_timing(True)
    eng.timing(reset=True)
    run(W + R * K, K)
    sync()
    tm = eng.timing()

    # sanity: the field must still be finite
    if not bool(torch.isfinite(runner.st.grids[0][len(runner.st.grids[0]) // 2]).all()):
        raise SystemExit("bench: non-finite field")

    if rank == 0:
        if emu is not None:
            print(f"[emulated slab {emu[0]}/{emu[1]}: {loc.Nx} planes, exchange = {args.emulate_transport}] "
                  f"{el / K * 1e3:.4f} ms/step -> {emu[1]} such ranks would give "
                  f"{sd.Npts * K / el / 1e9:.1f} Gvox/s if the exchange hides completely", file=sys.stderr)
        res = base_result(args, sd, world, K, W, R, regions, el, real_bytes, lossy, parallelism)
        rl, bpv, kernel_ms, units = roofline_block(args, sd, tm, K, real_bytes, interior_planes)
        res["roofline"] = rl
        if world > 1:
            res["exchange_verified"] = runner.exchange_verified
            res["exchange"] = {"backend": backend, "transport": runner.exchange_backend, "ranks": world, "checked_steps": min(W, 6),
                               "what": "bit-pattern checksums of the received ghost planes == the senders' planes, all ranks"}
        if single and not args.no_selfcheck:
            # Did the timed run compute the right thing?  The same steps from the same seeded field through ANOTHER kernel
            # family (single steps instead of whatever the engine chose -- blocked pairs at this size): every receiver sample
            # of every step must agree bit for bit (the families share no interior kernel; each is pinned to the CPU oracle
            # at small sizes by tests/).
            u_first = loc.u_out.copy()  # (the runner's engine writes its slab's receiver rows: for one slab, all of them)
            runner.st.close()
            torch.cuda.empty_cache()
            # 0x4000: single steps only; 0x8000: no creation-time measurement (static rules pick the single-step kernel)
            ekw2 = dict(ekw, debug=args.debug | 0x4000 | 0x8000)
            t0 = time.perf_counter()
            r2, loc2, _ = pdist.make_hip_runner(sd, 0, 1, local_rank, None, **ekw2)
            fill(r2.st, 1234 + rank)
            r2.st.eng.run(0, nt_total)
            r2.st.eng.sync()
            tm2 = r2.st.eng.timing()
            same = bool(np.array_equal(loc2.u_out, u_first))
            res["selfcheck"] = {"family_agreement": same, "steps": int(nt_total), "receiver_samples": int(u_first.size),
                                "max_abs_sample": float(np.abs(u_first).max()),
                                "timed_family": FAMILY.get(tm.get("air_path"), "other"),
                                "check_family": FAMILY.get(tm2.get("air_path"), "other"),
                                "seconds": round(time.perf_counter() - t0, 2)}
            r2.st.close()
            torch.cuda.empty_cache()
            if not same or not np.abs(u_first).max() > 0:
                print(json.dumps(res), flush=True)
                raise SystemExit("bench: the timed run's receivers differ from the check family's (or are all zero)")
        elif world == 1:
            runner.st.close()
            torch.cuda.empty_cache()
        if world == 1 and lossy and not args.no_rigid_run and emu is None:
            # a rigid-wall run next to the frequency-dependent one (same grid, no branch ODEs)
            sd_r = build_scene(n, 3 * K + W, args.precision, args.fcc, False, args.mb, args.nx, args.ny)
            rr, loc_r, _ = pdist.make_hip_runner(sd_r, 0, 1, local_rank, None, **ekw)
            fill(rr.st, 99)
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
                                  "blocked_pairs": tm_r.get("air_path") == 2,
                                  "autotune_ms_per_step": {k: round(v, 4) for k, v in zip(("lean", "barrier_free", "blocked_pair"), tm_r.get("tune_ms", [0, 0, 0]))}}
            rr.st.close()
        if world == 1 and emu is None:
            # HBM traffic of the dominant kernel: measured now (every engine of this process is closed, the device is free), on
            # this box, by counter passes of this same command; the committed passes only when rocprofv3 cannot run here
            live = None if args.no_pmc else measure_traffic_live(args, rl["kernel_instantiation"])
            add_traffic(rl, res, sd, kernel_ms, units, bpv, live)
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args.precision, args.fcc, args.mb, lossy)
        print(json.dumps(res), flush=True)

    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        runner.close_comm()  # (the library's own communicator, if the exchange ran on one)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
