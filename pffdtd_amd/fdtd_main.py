"""Drop-in for the reference binaries `fdtd_main_{cpu,gpu}_{single,double}.x` (c_cuda/fdtd_main.c:35-59).

    cd <sim_data folder> && python -m pffdtd_amd.fdtd_main --precision single [--gpus N]

Same flow: load_sim_data -> scale_input -> run_sim -> rescale_output -> write_outputs -> print_last_samples,
reading the four input .h5 files from the current directory and writing sim_outs.h5 there.

`--gpus N` is the counterpart of the reference GPU binary driving every visible device from one process
(gpu_engine.h:680-690, 993-1145): ONE process, the C library's chain object (pf_multi_*, csrc/pf_multi.hip: the grid cut
into Z-slabs like gpu_engine.h:516-662, one host thread per slab, ghost planes by peer copies or native RCCL on the edge
stream) on devices 0 .. N-1 -- round 4: the one N > 1 driver of the product.  `--devices 0,1,2,3` names the chain
explicitly; a device id may repeat (several slabs on one GPU).  Unlike the reference (gpu_engine.h:688) the lists need
not be sorted: the slabs are cut by plane tests and every engine sorts its own lists.

Under an external launcher (`python -m torch.distributed.run --nproc-per-node N -m pffdtd_amd.fdtd_main ...`: one process
per GPU, WORLD_SIZE set) each rank cuts its own slab (`pffdtd_amd/slab.py`), steps it with the split-phase engine and
exchanges over RCCL through torch.distributed; rank 0 gathers the receiver rows and writes `sim_outs.h5`.

`--progress [K]` prints the reference's progress fields (fdtd_common.h:106-190: T / I / TPW / IPW / TA / IA / TB / IB Mvox/s
and the air share, "I" = over the last K steps, default 200) as one plain line per report -- the reference redraws six
lines per STEP, which would cost a device synchronisation per step here.
"""
import argparse
import os
import time
from pathlib import Path

from . import engine, sim_data


def _summary(sd, tm, el):
    t_air = tm["air_ms_total"] * 1e-3
    t_rest = max(tm["step_ms_total"] * 1e-3 - t_air, 0.0)
    # the reference's three summary lines (cpu_engine.h:355-357 / gpu_engine.h:1251-1253), HIP-event timed
    print(f"Air update: {t_air:.6f}s, {sd.Npts * sd.Nt / 1e6 / max(t_air, 1e-12):.2f} Mvox/s")
    print(f"Boundary loop: {t_rest:.6f}s, {sd.Nb * sd.Nt / 1e6 / max(t_rest, 1e-12):.2f} Mvox/s")
    print(f"Combined (total): {el:.6f}s, {sd.Npts * sd.Nt / 1e6 / el:.2f} Mvox/s")


def _finish(sd, data_dir):
    sd.rescale_output()
    sd.write_outputs(data_dir)
    print("wrote output dataset")
    sd.print_last_samples(5)
    print(f"--Date and time: {time.ctime()}")


def _progress(sd, n, nt, t_all, t_chunk, k, air_all, air_chunk, bn_all, bn_chunk, workers):
    """one report in the reference's vocabulary (fdtd_common.h:106-190)"""
    def hms(sec):
        sec = int(sec)
        return f"{sec // 3600:02d}:{sec % 3600 // 60:02d}:{sec % 60:02d}"
    mv = lambda cells, t: 1e-6 * cells / max(t, 1e-12)
    print(f"Running [{100.0 * n / nt:.1f}%] [{hms(t_all)}<{hms(t_all * nt / max(n, 1))}] "
          f"T: {mv(sd.Npts * n, t_all):06.1f} - I: {mv(sd.Npts * k, t_chunk):06.1f} | "
          f"TPW: {mv(sd.Npts * n, t_all) / workers:06.1f} - IPW: {mv(sd.Npts * k, t_chunk) / workers:06.1f} | "
          f"TA: {mv(sd.Npts * n, air_all):06.1f} - IA: {mv(sd.Npts * k, air_chunk):06.1f} | "
          f"TB: {mv(sd.Nb * n, bn_all):06.1f} - IB: {mv(sd.Nb * k, bn_chunk):06.1f} | "
          f"T: {100.0 * air_all / max(t_all, 1e-12):02.1f}% - I: {100.0 * air_chunk / max(t_chunk, 1e-12):02.1f}%", flush=True)


def _run_chain(a, sd, m):
    """all Nt steps of a chain object, in one go or in chunks with progress reports; returns (seconds, timing of the busiest slab)"""
    live = range(m.nslabs)
    t0 = time.perf_counter()
    if not a.progress:
        m.run(0, sd.Nt)
        el = time.perf_counter() - t0
        tms = [m.slab(g)["engine"].timing() for g in live]
        return el, {"air_ms_total": max(t["air_ms_total"] for t in tms), "step_ms_total": max(t["step_ms_total"] for t in tms)}
    air_all = step_all = 0.0
    n = 0
    while n < sd.Nt:
        k = min(a.progress, sd.Nt - n)
        tc = time.perf_counter()
        m.run(n, k)
        now = time.perf_counter()
        n += k
        tms = [m.slab(g)["engine"].timing(reset=True) for g in live]
        air = max(t["air_ms_total"] for t in tms) * 1e-3
        step = max(t["step_ms_total"] for t in tms) * 1e-3
        if step <= 0:
            step = now - tc  # (slab engines time their interior launches only)
        air_all += air
        step_all += step
        _progress(sd, n, sd.Nt, now - t0, now - tc, k, air_all, air, max(step_all - air_all, 1e-12), max(step - air, 1e-12), m.nslabs)
    return time.perf_counter() - t0, {"air_ms_total": air_all * 1e3, "step_ms_total": step_all * 1e3}


def run_single(a):
    """One device.  Rooms (scenes the library stores with the x and z axes exchanged) run as TWO slabs on it, like pf_run_sim:
    the halves' kernels overlap (CTK church 313 against 288 Gvox/s); box rooms as one domain."""
    print(f"--Date and time: {time.ctime()}")
    sd = sim_data.SimData.from_folder(Path(a.data_dir), a.precision, build_mask=False)
    sd.scale_input()
    from .dist import scene_prefers_exchanged_axes
    two = sd.Nz >= 64 and scene_prefers_exchanged_axes(sd)  # (`--devices 0` = one domain on device 0)
    m = engine.HipMulti(sd, [a.gpu] * (2 if two else 1), timing=1)
    if two:
        print(f"--2 slabs on device {a.gpu}, cut along file z: {[(m.slab(g)['x0'], m.slab(g)['x1']) for g in range(2)]}")
    # sub-timers: the busier slab's HIP-event sums (two slabs run side by side on the device)
    el, tm = _run_chain(a, sd, m)
    if tm["step_ms_total"] <= 0:  # (slab engines time their interior launches only)
        tm["step_ms_total"] = el * 1e3
    m.close()
    _summary(sd, tm, el)
    _finish(sd, a.data_dir)


def run_devices(a, devices):
    """`--gpus N` / `--devices`: the C library's chain object in this process (what pf_run_sim_devices does inside)"""
    print(f"--Date and time: {time.ctime()}")
    sd = sim_data.SimData.from_folder(Path(a.data_dir), a.precision, build_mask=False)
    sd.scale_input()
    try:  # (the library checks the slab count against the axis it actually cuts -- file x, or file z for rooms; gpu_engine.h:682)
        m = engine.HipMulti(sd, devices, timing=1, verify_exchange=2)
    except engine.PfError as e:
        raise SystemExit(f"cannot run on {len(devices)} slabs: {e}")
    info = m.info()
    spans = [(m.slab(g)["x0"], m.slab(g)["x1"]) for g in range(m.nslabs)]
    print(f"--{len(devices)} slabs on devices {devices}, cut along file {'z' if info['cut_along_z'] else 'x'}: planes {spans}, ghost planes by {info['transport_name']}")
    el, tm = _run_chain(a, sd, m)
    if tm["step_ms_total"] <= 0:
        tm["step_ms_total"] = el * 1e3
    info = m.info()
    m.close()
    if info["exchange_verified"] is False:
        raise SystemExit("slab exchange self-check failed: a ghost plane does not hold what the neighbour sent")
    _summary(sd, tm, el)
    _finish(sd, a.data_dir)


def run_rank(a, world):
    """One rank of a `--gpus N` run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from torch.distributed.run)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL intra-node transport on this driver (dmabuf IPC)
    import torch
    import torch.distributed as dist
    from . import dist as pdist
    rank, local_rank = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    # debug only: PFFDTD_BACKEND=gloo runs every rank on GPU 0 with host-staged planes (one-GPU boxes)
    backend = os.environ.get("PFFDTD_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = a.gpu
    if engine.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: no GPU {local_rank} (the HIP engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "gloo":
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    try:
        if rank == 0:
            print(f"--Date and time: {time.ctime()}")
        sd = sim_data.SimData.from_folder(Path(a.data_dir), a.precision, build_mask=False)
        if world >= sd.Nx:
            raise SystemExit(f"need ngpus < Nx (got {world}, Nx={sd.Nx})")  # gpu_engine.h:682
        sd.scale_input()
        runner, loc, info = pdist.make_hip_runner(sd, rank, world, local_rank, None, timing=True)
        if rank == 0:
            az = bool(getattr(info, "along_z", False))
            print(f"--{world} GPUs, slabs of {[x1 - x0 for x0, x1 in pdist.slab_mod.partition_weighted(sd, world, az, getattr(info, 'wall_scale', 1.0))]} planes"
                  + (" cut along file z (engines store the x and z axes exchanged)" if az else " along x"))
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run(0, sd.Nt)
        runner.finish()
        dist.barrier()
        el = time.perf_counter() - t0
        tm = runner.st.eng.timing()
        pdist.gather_outputs(sd, loc, info)
        runner.close_comm()
        runner.st.close()
        if rank == 0:
            _summary(sd, tm, el)  # air / boundary split: rank 0's slab; the total: the whole job
            _finish(sd, a.data_dir)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--precision", default="single", choices=["single", "double"])
    p.add_argument("--data_dir", default=".", help="folder with the input .h5 files (the reference uses the CWD)")
    p.add_argument("--gpu", type=int, default=0, help="device of a single-GPU run")
    p.add_argument("--gpus", type=int, default=1, help="number of GPUs: Z-slabs on devices 0 .. N-1, driven from this one process")
    p.add_argument("--devices", default="", help="comma-separated device chain instead of 0 .. N-1, e.g. 0,1,2,3 (ids may repeat)")
    p.add_argument("--progress", type=int, nargs="?", const=200, default=0, metavar="K",
                   help="report the reference's progress fields (fdtd_common.h:106-190) every K steps (default 200)")
    a = p.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.devices:
        return run_devices(a, [int(v) for v in a.devices.split(",")])
    if world > 1:
        if a.gpus not in (1, world):
            raise SystemExit(f"--gpus {a.gpus} != WORLD_SIZE {world}")
        return run_rank(a, world)
    if a.gpus > 1:
        ndev = engine.device_count()
        if ndev < a.gpus:
            raise SystemExit(f"--gpus {a.gpus}: only {ndev} device(s) visible (name a chain with repeats by --devices for virtual slabs)")
        return run_devices(a, list(range(a.gpus)))
    run_single(a)


if __name__ == "__main__":
    main()
