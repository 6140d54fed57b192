"""GPU: the real multi-process slab runner (one process per rank, torch.distributed) with every rank on GPU 0 and the
planes staged through gloo -- the whole N>1 control flow (slab split, HIP split-phase stepping, exchange ordering,
output gather) except the RCCL transport itself, which needs one GPU per rank.  Bit-exact vs the oracle."""
import os

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
import oracle

pytestmark = pytest.mark.gpu


def _big_sd(prec):
    """A box room wide enough for the temporally blocked pairs of the slab engines (>= 248 columns in the box)."""
    from pffdtd_amd import sim_data, synth
    sim = synth.shoebox(96, 64, 280, Nt=19, Nm=2, Mb=[11, 3], rcv=[[50, 30, 140], [7, 8, 9], [88, 55, 260]])
    sd = sim_data.SimData.from_sim(sim, prec)
    sd.scale_input()
    return sd


def _worker(rank, world, port, name, prec, q, along_z=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pffdtd_amd import dist as pdist
        paired = name == "box_pairs"
        sd = _big_sd(prec) if paired else cases.make_sd(name, prec)
        runner, loc, info = pdist.make_hip_runner(sd, rank, world, 0, along_z=along_z, **(dict(air_variant=40, pairs=True) if paired else {}))
        assert bool(runner.st.paired) == paired and bool(getattr(info, "along_z", False)) == along_z
        if along_z:
            assert runner.st.eng.layout()[2] is True and len(runner.st.grids[0]) == loc.Nz
        runner.run(0, sd.Nt)
        runner.finish()
        out = pdist.gather_outputs(sd, loc, info)
        runner.st.close()
        if rank == 0:
            q.put(out.copy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name,prec,along_z", [(2, "cart_outside", "single", False), (3, "fcc2_outside", "double", False),
                                                     (2, "box_pairs", "single", False), (3, "box_pairs", "single", False),
                                                     (2, "cart_outside", "single", True), (3, "fcc2_outside", "double", True),
                                                     (2, "fcc1_outside", "single", True)])
def test_multiprocess_hip_slabs(world, name, prec, along_z):
    """(along_z: the chain cut along FILE Z, every rank's engine storing the x and z axes exchanged -- what rooms get by default)"""
    sd = _big_sd(prec) if name == "box_pairs" else cases.make_sd(name, prec)
    oracle.run_sim(sd)
    ref = sd.u_out.copy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, prec, q, along_z)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(out, ref)


def test_bench_two_ranks_control_flow():
    """bench.py as the driver launches it for N>1 (torch.distributed.run, one process per rank), here with both ranks
    on GPU 0 and gloo-staged planes (PFFDTD_BENCH_BACKEND=gloo): one JSON line from rank 0 with the contract's keys."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, PFFDTD_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", str(root / "bench.py"), "--gpus", "2", "--size", "320", "--steps", "6", "--warmup", "3",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 6 and res["warmup"] == 3 and res["value"] > 0
    assert res["metric"] == "Gvoxel-updates/s" and res["scaling"] == "strong" and res["roofline"]["bound"] == "hbm"
    # the exchange self-check ran during warm-up on both ranks and found the received ghost planes equal to the sent ones
    assert res["exchange_verified"] is True and res["exchange"]["ranks"] == 2 and res["repeats"] >= 1
