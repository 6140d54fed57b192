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


def _worker(rank, world, port, name, prec, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pffdtd_amd import dist as pdist
        paired = name == "box_pairs"
        sd = _big_sd(prec) if paired else cases.make_sd(name, prec)
        runner, loc, info = pdist.make_hip_runner(sd, rank, world, 0, **(dict(air_variant=40, pairs=True) if paired else {}))
        assert runner.st.paired == paired
        runner.run(0, sd.Nt)
        runner.finish()
        out = pdist.gather_outputs(sd, loc, info)
        runner.st.close()
        if rank == 0:
            q.put(out.copy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name,prec", [(2, "cart_outside", "single"), (3, "fcc2_outside", "double"),
                                             (2, "box_pairs", "single"), (3, "box_pairs", "single")])
def test_multiprocess_hip_slabs(world, name, prec):
    sd = _big_sd(prec) if name == "box_pairs" else cases.make_sd(name, prec)
    oracle.run_sim(sd)
    ref = sd.u_out.copy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, prec, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(out, ref)
