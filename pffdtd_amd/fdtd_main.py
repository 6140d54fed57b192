"""Drop-in for the reference binaries `fdtd_main_{cpu,gpu}_{single,double}.x` (c_cuda/fdtd_main.c:35-59).

    cd <sim_data folder> && python -m pffdtd_amd.fdtd_main --precision single [--gpus N]

Same flow: load_sim_data -> scale_input -> run_sim -> rescale_output -> write_outputs -> print_last_samples,
reading the four input .h5 files from the current directory and writing sim_outs.h5 there.

`--gpus N` is the counterpart of the reference GPU binary driving every visible device from one thread
(gpu_engine.h:680-690, 993-1145): here the command re-launches itself as N processes (one per GPU,
`torch.distributed.run`), each cuts its Z-slab out of the folder's lists (`pffdtd_amd/slab.py` =
gpu_engine.h:516-662), steps it with the split-phase engine and exchanges one plane per side and step over RCCL;
rank 0 gathers the receiver rows and writes `sim_outs.h5`.  Unlike the reference (gpu_engine.h:688) the lists need
not be sorted: the slabs are cut by plane tests and every engine sorts its own lists.
It can also be started under `python -m torch.distributed.run --nproc-per-node N -m pffdtd_amd.fdtd_main ...` directly.

`--devices 0,1,2,3` is the in-process alternative, closest to the reference binary: ONE process, the C library's own
multi-device `run_sim` (pf_run_sim_devices, csrc/pf_multi.hip: one host thread per slab, ghost planes pulled with peer
copies), no torch and no RCCL involved; a device id may repeat (several slabs on one GPU).
"""
import argparse
import os
import subprocess
import sys
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


def run_single(a):
    """One device.  Rooms (scenes the library stores with the x and z axes exchanged) run as TWO slabs on it, like pf_run_sim:
    the halves' kernels overlap (CTK church 313 against 288 Gvox/s); box rooms as one domain."""
    print(f"--Date and time: {time.ctime()}")
    sd = sim_data.SimData.from_folder(Path(a.data_dir), a.precision, build_mask=False)
    sd.scale_input()
    from .dist import scene_prefers_exchanged_axes
    two = sd.Nz >= 64 and os.environ.get("PFFDTD_SLABS_PER_DEVICE", "") != "1" and scene_prefers_exchanged_axes(sd)
    m = engine.HipMulti(sd, [a.gpu] * (2 if two else 1), timing=1)
    t0 = time.perf_counter()
    m.run(0, sd.Nt)
    el = time.perf_counter() - t0
    tms = [m.slab(g)["engine"].timing() for g in range(m.nslabs)]
    if two:
        print(f"--2 slabs on device {a.gpu}, cut along file z: {[(m.slab(g)['x0'], m.slab(g)['x1']) for g in range(2)]}")
    # sub-timers: the busier slab's HIP-event sums (two slabs run side by side on the device)
    tm = {"air_ms_total": max(t["air_ms_total"] for t in tms), "step_ms_total": max(t["step_ms_total"] for t in tms)}
    if tm["step_ms_total"] <= 0:  # (slab engines time their interior launches only)
        tm["step_ms_total"] = el * 1e3
    m.close()
    _summary(sd, tm, el)
    _finish(sd, a.data_dir)


def run_devices(a, devices):
    """`--devices`: the C seam's multi-device run_sim in this process"""
    print(f"--Date and time: {time.ctime()}")
    sd = sim_data.SimData.from_folder(Path(a.data_dir), a.precision, build_mask=False)
    sd.scale_input()
    print(f"--{len(devices)} slabs on devices {devices}: planes {engine.slab_partition(sd, len(devices))}")
    t0 = time.perf_counter()
    engine.run_sim_devices(sd, devices)
    el = time.perf_counter() - t0
    print(f"Combined (total): {el:.6f}s, {sd.Npts * sd.Nt / 1e6 / el:.2f} Mvox/s")  # incl. engine creation on every device
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
            print(f"--{world} GPUs, slabs of {[x1 - x0 for x0, x1 in pdist.slab_mod.partition_weighted(sd, world, az)]} planes"
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
    p.add_argument("--gpus", type=int, default=1, help="number of GPUs (one process each, Z-slabs)")
    p.add_argument("--master_port", type=int, default=29541)
    p.add_argument("--devices", default="", help="comma-separated device chain for the in-process multi-device run, e.g. 0,1,2,3")
    a = p.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.devices:
        return run_devices(a, [int(v) for v in a.devices.split(",")])
    if world > 1:
        if a.gpus not in (1, world):
            raise SystemExit(f"--gpus {a.gpus} != WORLD_SIZE {world}")
        return run_rank(a, world)
    if a.gpus > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(a.master_port), "-m", "pffdtd_amd.fdtd_main",
               "--precision", a.precision, "--data_dir", str(Path(a.data_dir).resolve()), "--gpus", str(a.gpus),
               "--gpu", str(a.gpu)]
        env = dict(os.environ)
        root = str(Path(__file__).resolve().parent.parent)
        env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        raise SystemExit(subprocess.call(cmd, env=env))
    run_single(a)


if __name__ == "__main__":
    main()
