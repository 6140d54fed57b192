"""Host-side cost of the split-phase slab loop (debug): per-step time spent in step_begin / exchange / step_end
and whether the host runs ahead of the GPU.  usage: python tools/hostprof.py [copy|rccl] [r/N]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import torch.distributed as dist
from pffdtd_amd import synth, sim_data, dist as pdist

mode = sys.argv[1] if len(sys.argv) > 1 else "copy"
r, N = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "3/8").split("/"))
n = 1024
sim = synth.shoebox(n, n, n, Nt=1400, Nm=1, Mb=11, lossy=True)
sd = sim_data.SimData.from_sim(sim, "single", build_mask=False)
sd.scale_input()
if mode != "copy":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29733")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
TIMING = os.environ.get("HP_TIMING", "0") == "1"
runner, loc, info = pdist.make_hip_runner(sd, r, N, 0, None, timing=TIMING)
st = runner.st
if os.environ.get("HP_RAND", "0") == "1":
    for g in st.grids:
        g.copy_((torch.rand(g.shape, device=g.device, dtype=torch.float32) * 2.0 - 1.0) * 1e-3)
    torch.cuda.synchronize()


def ex_copy():
    s_lo, s_hi, r_lo, r_hi = st.halo_tensors()
    with st.comm_context():
        r_lo.copy_(s_lo, non_blocking=True)
        r_hi.copy_(s_hi, non_blocking=True)


def ex_rccl():
    s_lo, s_hi, r_lo, r_hi = st.halo_tensors()
    ops = [dist.P2POp(dist.isend, s_lo, 0), dist.P2POp(dist.irecv, r_lo, 0),
           dist.P2POp(dist.isend, s_hi, 0), dist.P2POp(dist.irecv, r_hi, 0)]
    with st.comm_context():
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def ex_rccl_plain():
    s_lo, s_hi, r_lo, r_hi = st.halo_tensors()
    with st.comm_context():
        w = [dist.isend(s_lo, 0), dist.irecv(r_lo, 0)]
        for x in w:
            x.wait()


ex = {"copy": ex_copy, "rccl": ex_rccl, "plain": ex_rccl_plain}[mode]
for rep in range(3):
    st.sync()
    torch.cuda.synchronize()
    tb = te = tx = 0.0
    t00 = time.perf_counter()
    for k in range(100):
        nn = rep * 100 + k
        t0 = time.perf_counter()
        st.step_begin(nn)
        t1 = time.perf_counter()
        ex()
        t2 = time.perf_counter()
        st.step_end(nn)
        t3 = time.perf_counter()
        tb += t1 - t0
        tx += t2 - t1
        te += t3 - t2
    tenq = time.perf_counter() - t00
    st.sync()
    torch.cuda.synchronize()
    ttot = time.perf_counter() - t00
    print(f"{mode} {r}/{N}: host per step: begin {tb*1e4:.1f} us  exchange {tx*1e4:.1f} us  end {te*1e4:.1f} us; "
          f"enqueue {tenq*1e4:.1f} total {ttot*1e4:.1f} us")
# one long unsynchronised loop: does the host keep running ahead?
st.sync()
t00 = time.perf_counter()
marks = []
for k in range(900):
    st.step_begin(300 + k)
    ex()
    st.step_end(300 + k)
    if k % 100 == 99:
        marks.append((time.perf_counter() - t00) * 1e3)
st.sync()
print(f"{mode} timing={TIMING}: host clock after every 100 steps (ms): " + " ".join(f"{m:.2f}" for m in marks) + f"; all done {(time.perf_counter() - t00) * 1e3:.2f}")
st.close()
if mode != "copy":
    dist.destroy_process_group()
