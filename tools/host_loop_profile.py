#!/usr/bin/env python3
"""Where does the host time of one split-phase step go in the one-process-per-GPU loop (pffdtd_amd/dist.py)?  Rank r of N alone on the
device, exchanging with itself through torch.distributed's RCCL p2p (world_size 1), every piece of the loop timed on the host.
usage (GPU box): python tools/host_loop_profile.py [r/N] [steps]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from pffdtd_amd import dist as pdist  # noqa: E402

spec = sys.argv[1] if len(sys.argv) > 1 else "3/8"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 120
r, N = (int(v) for v in spec.split("/"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29741")
torch.cuda.set_device(0)
opts = None
if os.environ.get("HIPRIO", "1") == "1":  # RCCL's own stream at high priority, like the engine's edge stream (HIPRIO=0: torch's default)
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=opts)
sd = bench.build_scene(1024, K + 30, "single", False, True, 11, 0, 0)
runner, loc, info = pdist.make_hip_runner(sd, r, N, 0, None, numerics=0, air_variant=0, air_chunk=0, timing=False, debug=0)
st = runner.st
NATIVE = os.environ.get("NATIVE", "0") == "1"  # the library's own ncclSend / ncclRecv on the edge stream (what dist.py uses under torchrun)
if NATIVE and not runner.enable_native_rccl(0, peers=(-1 if info.first else 0, -1 if info.last else 0)):
    raise SystemExit("native RCCL unavailable: " + getattr(runner, "native_note", "?"))
T = {k: 0.0 for k in ("step_begin", "halo_tensors", "p2pops", "batch_isend_irecv", "wait", "step_end")}


def step(n, rec):
    t0 = time.perf_counter(); st.step_begin(n)
    if NATIVE:
        t1 = time.perf_counter(); runner.exchange()
        t5 = time.perf_counter(); st.step_end(n)
        t6 = time.perf_counter()
        if rec:
            T["step_begin"] += t1 - t0; T["batch_isend_irecv"] += t5 - t1; T["step_end"] += t6 - t5
        return
    t1 = time.perf_counter(); s_lo, s_hi, r_lo, r_hi = st.halo_tensors()
    t2 = time.perf_counter()
    ops = []
    if not info.first:
        ops += [dist.P2POp(dist.isend, s_lo, 0), dist.P2POp(dist.irecv, r_lo, 0)]
    if not info.last:
        ops += [dist.P2POp(dist.isend, s_hi, 0), dist.P2POp(dist.irecv, r_hi, 0)]
    t3 = time.perf_counter()
    with st.comm_context():
        works = dist.batch_isend_irecv(ops)
        t4 = time.perf_counter()
        for w in works:
            w.wait()
    t5 = time.perf_counter(); st.step_end(n)
    t6 = time.perf_counter()
    if rec:
        for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
            T[k] += v


for n in range(30):
    step(n, False)
st.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
for n in range(30, 30 + K):
    step(n, True)
host = time.perf_counter() - t0
st.sync(); torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"rank {r}/{N}: {K} steps, host loop {host / K * 1e3:.4f} ms/step, with the device drained {wall / K * 1e3:.4f} ms/step")
for k, v in T.items():
    print(f"   {k:20s} {v / K * 1e6:8.1f} us/step")
