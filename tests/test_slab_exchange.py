"""The Z-slab time loop (pffdtd_amd/slab.py + pffdtd_amd/dist.py) on CPU: world_size-2/3 gloo processes, each
slab stepped by the CPU ORACLE (test infrastructure standing in for the HIP stepper), must reproduce the
single-domain oracle bit for bit.  This covers the partition, the index re-basing, the exchange schedule and the
output merge -- everything of the N>1 path except the HIP kernels themselves (covered by -m gpu tests).
"""
import contextlib
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
import oracle
from pffdtd_amd import dist as pdist
from pffdtd_amd import slab


class OracleSlabStepper:
    """Same interface as pffdtd_amd.dist.HipSlabStepper, backed by oracle.Engine (CPU, tests only)."""

    def __init__(self, loc, info):
        self.loc, self.info = loc, info
        self.zcut = bool(getattr(info, "along_z", False))  # a chain cut along file z: the exchanged "planes" are z columns
        self.e = oracle.Engine(loc, slab_first=info.first, slab_last=info.last, along_z=self.zcut)
        self._staged = None

    def step_begin(self, n):
        self.e.step(n)  # whole step; the new state is u1 after the rotation
        self._staged = None

    def halo_tensors(self):
        g = self.e.grid(1)
        if self.zcut:  # strided in the file layout: staged through contiguous buffers, written back in step_end
            if self._staged is None:
                Nz = self.loc.Nz
                c = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1))  # noqa: E731
                self._staged = (c(g[:, :, 1]), c(g[:, :, Nz - 2]), c(g[:, :, 0]), c(g[:, :, Nz - 1]))
            return self._staged
        Nx = self.loc.Nx
        f = lambda a: torch.from_numpy(a.reshape(-1))  # noqa: E731  (views: irecv writes in place)
        return f(g[1]), f(g[Nx - 2]), f(g[0]), f(g[Nx - 1])

    def comm_context(self):
        return contextlib.nullcontext()

    def step_end(self, n):
        if self.zcut and self._staged is not None:
            g = self.e.grid(1)
            if not self.info.first:
                g[:, :, 0] = self._staged[2].numpy().reshape(g.shape[0], g.shape[1])
            if not self.info.last:
                g[:, :, -1] = self._staged[3].numpy().reshape(g.shape[0], g.shape[1])

    def finish(self):
        pass


def _reference(name, prec):
    sd = cases.make_sd(name, prec)
    oracle.run_sim(sd)
    return sd.u_out.copy()


@pytest.mark.parametrize("G", [2, 3])
@pytest.mark.parametrize("name,prec", [("cart_outside", "single"), ("fcc2_outside", "double"), ("cart_lossy", "double")])
@pytest.mark.parametrize("along_z", [False, True])
def test_in_process_slabs_equal_single_domain(name, prec, G, along_z):
    """G slabs in one process, planes copied directly: isolates slab.split from the transport.  along_z: the chain cut along
    FILE Z (what rooms get: their engines store the x and z axes exchanged)."""
    ref = _reference(name, prec)
    sd = cases.make_sd(name, prec)
    parts = [slab.split(sd, G, r, balance=(G == 3), along_z=along_z) for r in range(G)]
    st = [OracleSlabStepper(loc, info) for loc, info in parts]
    for n in range(sd.Nt):
        for s in st:
            s.step_begin(n)
        planes = [s.halo_tensors() for s in st]
        for r in range(G - 1):
            planes[r + 1][2].copy_(planes[r][1])      # my last updated plane -> right neighbour's low ghost
            planes[r][3].copy_(planes[r + 1][0])      # right neighbour's first updated plane -> my high ghost
        for s in st:
            s.step_end(n)
    out = slab.merge_outputs(sd, [p[0] for p in parts])
    assert np.array_equal(out, ref)


def _worker(rank, world, port, name, prec, q, along_z=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        oracle.lib().oracle_set_threads(2)
        sd = cases.make_sd(name, prec)
        loc, info = slab.split(sd, world, rank, along_z=along_z)
        runner = pdist.SlabRunner(OracleSlabStepper(loc, info), info)
        runner.verify_steps = sd.Nt  # checksum every exchange against the senders' planes (bench.py does a few)
        runner.run(0, sd.Nt)
        runner.finish()
        verified = runner.exchange_verified
        # the check must also be able to FAIL: corrupt one received plane on the last rank and look again
        if rank == world - 1:
            runner.st.halo_tensors()[2][3] += 1.0
        runner._verify_exchange()
        caught = runner.exchange_verified is False
        out = pdist.gather_outputs(sd, loc, info)
        if rank == 0:
            q.put((out.copy(), verified, caught))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("along_z", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_gloo_slab_chain_equals_single_domain(world, along_z):
    name, prec = "cart_outside", "single"
    ref = _reference(name, prec)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world + (10 if along_z else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, prec, q, along_z)) for r in range(world)]
    for p in procs:
        p.start()
    out, verified, caught = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(out, ref)
    assert verified is True and caught is True
