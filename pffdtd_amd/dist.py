"""Multi-GPU time loop: one process per GPU, Z-slab chain, one-plane ghost exchange per step.

Counterpart of the multi-GPU part of the reference `run_sim` (c_cuda/gpu_engine.h:993-1145), re-designed:
the reference drives all GPUs from one host thread and does `cudaMemcpyPeerAsync` after synchronising
every stream (:1077-1126, "not async to rest of scheme"); here each rank owns one slab, the two edge
planes (+ every boundary-node list entry inside them) are computed first on a high-priority stream, the
exchange (`torch.distributed` P2P = RCCL send/recv over xGMI; gloo in the CPU tests) is ordered after
that stream only, and the interior planes run concurrently on the main stream.

The per-slab work is behind a small `stepper` interface so that the exchange schedule itself can be
tested on CPU with gloo (tests inject an oracle-backed stepper; the product stepper is HIP only).
"""
import os

import torch
import torch.distributed as dist

from . import slab as slab_mod


class _DevMem:
    """A device allocation that is not torch's, described for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class HipSlabStepper:
    """HIP engine of one slab whose state grids are addressable as tensors (halo planes, initial fields).  Slabs of a
    multi-rank run: torch-owned grids handed to the engine.  A single domain: the engine allocates -- it then also chooses
    where its four grids live (creation-time placement sampling) -- and `grids` are views of its allocations."""

    def __init__(self, loc, info, device, pairs=False, **engine_kw):
        from . import engine
        self.loc, self.info = loc, info
        self.device = torch.device("cuda", device)
        self.tdtype = torch.float32 if loc.real_bytes == 4 else torch.float64
        P = engine.grid_pitch(loc.Nz, loc.real_bytes)
        self.plane = loc.Ny * P
        if info.G == 1 and os.environ.get("PFFDTD_TORCH_GRIDS", "") != "1":
            self.eng = engine.HipEngine(loc, device=device, slab_first=info.first, slab_last=info.last, x_global0=info.xlo,
                                        **engine_kw)
            ts = "<f4" if loc.real_bytes == 4 else "<f8"
            (nx, ny, _), pitch, _ = self.eng.layout()  # as STORED: the engine may keep the file's x and z axes exchanged
            with torch.cuda.device(self.device):
                self.grids = [torch.as_tensor(_DevMem(p, (nx, ny * pitch), ts), device=self.device)
                              for p in self.eng.state_grids()]
            if [g.data_ptr() for g in self.grids] != list(self.eng.state_grids()):
                raise RuntimeError("torch copied the engine's state grids instead of wrapping them")
        else:
            # a slab of a chain cut along FILE Z: the engine stores planes of file z, Ny rows of pitch(Nx) each (PF_LAYOUT_EXCHANGED)
            zcut = bool(getattr(info, "along_z", False))
            if zcut:
                self.plane = loc.Ny * engine.grid_pitch(loc.Nx, loc.real_bytes)
                engine_kw = dict(engine_kw, layout=engine.PF_LAYOUT_EXCHANGED)
            self.nplanes = loc.Nz if zcut else loc.Nx
            with torch.cuda.device(self.device):
                self.grids = [torch.zeros((self.nplanes, self.plane), dtype=self.tdtype, device=self.device) for _ in range(2)]
                torch.cuda.synchronize()
            self.eng = engine.HipEngine(loc, device=device, slab_first=info.first, slab_last=info.last, x_global0=info.xlo,
                                        ext_u0=self.grids[0].data_ptr(), ext_u1=self.grids[1].data_ptr(), **engine_kw)
        # Two more grids let a slab engine with a boundary-free box advance in temporally blocked pairs (the state
        # then cycles through all four); engines that cannot use them say so and the grids are dropped again.
        # The pair kernel's speed depends on where its four grids lie relative to each other (grid placement):
        # the engine is offered a pool of up to eight and keeps the four it is fastest on.
        self.paired = False       # strictly a boolean: does the slab step in blocked pairs or triples?
        self.steps_per_pass = 0   # 0 single steps, 2 pairs, 3 triples
        if info.G > 1 and pairs:
            pool = list(self.grids)
            try:
                with torch.cuda.device(self.device):
                    for _ in range(4 + 2):
                        pool.append(torch.zeros((self.nplanes, self.plane), dtype=self.tdtype, device=self.device))
                    torch.cuda.synchronize()
            except torch.OutOfMemoryError:  # no room for more: what fits
                torch.cuda.empty_cache()
            if len(pool) >= 5:  # (five or more: the engine may step in triples, pf_engine_place_grids5)
                self.steps_per_pass, idx = self.eng.place_grids5([g.data_ptr() for g in pool])
                self.grids = [pool[i] for i in idx if i >= 0]
            elif len(pool) >= 4:
                two, idx = self.eng.place_grids([g.data_ptr() for g in pool])
                self.steps_per_pass = 2 if two else 0
                self.grids = [pool[i] for i in idx if i >= 0]
            self.paired = self.steps_per_pass > 0
            del pool
            torch.cuda.empty_cache()
        self._by_ptr = {g.data_ptr(): g for g in self.grids}
        self.edge_stream = torch.cuda.ExternalStream(self.eng.stream(1), device=self.device)
        self.main_stream = torch.cuda.ExternalStream(self.eng.stream(0), device=self.device)
        self.k = 0  # steps completed

    def step_begin(self, n):
        self.eng.step_begin(n)

    def halo_tensors(self):
        g = self._by_ptr[self.eng.halo_ptrs()[2]]  # the grid the step in flight writes (its plane 0 = recv_lo)
        Nx = len(g)                                # storage planes: file x planes, or file z planes of a chain cut along z
        return g[1], g[Nx - 2], g[0], g[Nx - 1]  # send_lo, send_hi, recv_lo, recv_hi

    def comm_context(self):
        """Exchange calls are issued with the edge stream current: RCCL orders itself after the edge planes only."""
        return torch.cuda.stream(self.edge_stream)

    def step_end(self, n):
        self.eng.step_end(n)
        self.k += 1

    def finish(self):
        self.eng.flush_outputs()
        self.eng.sync()

    def sync(self):
        self.eng.sync()

    def close(self):
        # a single domain's `grids` are views of the ENGINE's allocations: drop them before the engine frees the memory
        if self.grids and self.grids[0].is_cuda:
            torch.cuda.synchronize(self.device)
        self.grids = []
        self._by_ptr = {}
        self.eng.close()


class SlabRunner:
    """Time loop of one rank of the slab chain."""

    def __init__(self, stepper, info, group=None):
        self.st, self.info, self.group = stepper, info, group
        self.rank, self.G = info.rank, info.G
        self.verify_steps = 0          # >0: checksum the planes of that many upcoming exchanges against the senders'
        self.exchange_verified = None  # None = never checked, True / False = result on every rank (all-reduced)
        self.native = None             # pf_rccl_comm*: the planes by native ncclSend / ncclRecv on the engine's edge stream
        self.exchange_backend = "torch.distributed p2p"

    def enable_native_rccl(self, device, peers=None):
        """Collective over the group: every rank creates a communicator of the library's own (pf_rccl_*, include/pffdtd_hip.h) from an id
        rank 0 hands round, and the exchange becomes ONE ncclGroup of sends / receives on the engine's edge stream.  torch's own p2p
        runs on a stream of its own behind two cross-stream hops -- measured with a rank of 8 at 1024^3: 0.33 ms per step against 0.23-0.25
        (tools/host_loop_profile.py).  Any rank failing (no librccl, a rendezvous that errs or times out): all stay on torch's p2p.
        peers: (lo, hi) ranks of the communicator to exchange with instead of rank -/+ 1 (a rank alone exchanging with itself: (0, 0))."""
        import ctypes
        from . import engine
        if self.G == 1 and peers is None:
            return False
        L = engine.lib()
        idb = ctypes.create_string_buffer(128)
        nranks, rank = (self.G, self.rank) if peers is None else (1, 0)
        obj = [None]  # the id, or None if rank 0 could not get one
        if rank == 0 and L.pf_rccl_unique_id(idb) == 0:
            obj = [bytes(idb.raw)]
        together = nranks > 1
        if together:
            dist.broadcast_object_list(obj, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        comm = ctypes.c_void_p()
        good = obj[0] is not None and L.pf_rccl_comm_create(obj[0], nranks, rank, int(device), ctypes.byref(comm)) == 0
        flags = [good]
        if together:
            flags = [None] * nranks
            dist.all_gather_object(flags, bool(good), group=self.group)
        if not all(flags):
            if good:
                L.pf_rccl_comm_destroy(comm)
            self.native_note = engine.lib().pf_last_error().decode() if not good else "another rank could not create its communicator"
            return False
        self.native, self._peers = comm, peers
        self.exchange_backend = "native RCCL (ncclSend / ncclRecv grouped on the engine's edge stream)"
        return True

    def exchange(self):
        """Send my first/last updated planes to the neighbours' ghost planes, receive theirs (gpu_engine.h:1086-1126)."""
        if self.G == 1:
            return
        if self.native is not None:
            from . import engine
            lo, hi = self._peers if self._peers is not None else (-1 if self.info.first else self.rank - 1, -1 if self.info.last else self.rank + 1)
            if engine.lib().pf_rccl_exchange(self.native, self.st.eng._h, lo, hi) != 0:
                raise RuntimeError("pf_rccl_exchange: " + engine.lib().pf_last_error().decode())
            return
        s_lo, s_hi, r_lo, r_hi = self.st.halo_tensors()
        if s_lo.is_cuda and dist.get_backend(self.group) == "gloo":
            return self._exchange_staged(s_lo, s_hi, r_lo, r_hi)
        ops = []
        if not self.info.first:
            ops.append(dist.P2POp(dist.isend, s_lo, self.rank - 1, self.group))
            ops.append(dist.P2POp(dist.irecv, r_lo, self.rank - 1, self.group))
        if not self.info.last:
            ops.append(dist.P2POp(dist.isend, s_hi, self.rank + 1, self.group))
            ops.append(dist.P2POp(dist.irecv, r_hi, self.rank + 1, self.group))
        with self.st.comm_context():
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def _exchange_staged(self, s_lo, s_hi, r_lo, r_hi):
        """Debug transport (several ranks sharing ONE GPU, gloo): planes staged through host memory.  Exercises the
        whole multi-rank control flow where RCCL cannot run; never used on a real multi-GPU node."""
        with self.st.comm_context():
            torch.cuda.current_stream().synchronize()
            ops, backs = [], []
            for send, recv, peer, use in ((s_lo, r_lo, self.rank - 1, not self.info.first),
                                          (s_hi, r_hi, self.rank + 1, not self.info.last)):
                if use:
                    hbuf = torch.empty(recv.shape, dtype=recv.dtype)
                    ops.append(dist.P2POp(dist.isend, send.cpu(), peer, self.group))
                    ops.append(dist.P2POp(dist.irecv, hbuf, peer, self.group))
                    backs.append((recv, hbuf))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for recv, hbuf in backs:
                recv.copy_(hbuf)
            torch.cuda.current_stream().synchronize()

    def _verify_exchange(self):
        """Every rank checksums (bit patterns, exact) the two planes it sent and the two it received; the sums travel by
        all_gather and each rank checks that what arrived in its ghost planes is what its neighbours sent -- and that a
        send plane is not all zeros by accident of an unexercised path (the fields are pre-filled in bench.py)."""
        s_lo, s_hi, r_lo, r_hi = self.st.halo_tensors()
        with self.st.comm_context():
            if s_lo.is_cuda:
                torch.cuda.current_stream().synchronize()
            itype = torch.int32 if s_lo.element_size() == 4 else torch.int64

            def csum(t):
                return int(t.contiguous().view(itype).to(torch.int64).sum().item())
            mine = [csum(s_lo), csum(s_hi), csum(r_lo), csum(r_hi)]
        allv = [None] * self.G
        dist.all_gather_object(allv, mine, group=self.group)
        ok = True
        if not self.info.first:
            ok &= mine[2] == allv[self.rank - 1][1]   # my plane 0 = left neighbour's last owned plane
        if not self.info.last:
            ok &= mine[3] == allv[self.rank + 1][0]   # my last plane = right neighbour's first owned plane
        flags = [None] * self.G
        dist.all_gather_object(flags, bool(ok), group=self.group)
        good = all(flags)
        self.exchange_verified = good if self.exchange_verified is None else (self.exchange_verified and good)

    def run(self, n0, nsteps):
        for n in range(n0, n0 + nsteps):
            self.st.step_begin(n)
            self.exchange()
            if self.verify_steps > 0 and self.G > 1:
                self.verify_steps -= 1
                self._verify_exchange()
            self.st.step_end(n)

    def finish(self):
        self.st.finish()

    def close_comm(self):
        if self.native is not None:
            from . import engine
            self.st.sync()
            engine.lib().pf_rccl_comm_destroy(self.native)
            self.native = None


def gather_outputs(sd, loc, info, group=None):
    """Collect every slab's receiver rows on all ranks (small: Nr x Nt doubles)."""
    if info.G == 1:
        sd.u_out[loc.out_rows, :] = loc.u_out
        return sd.u_out
    parts = [None] * info.G
    dist.all_gather_object(parts, (loc.out_rows, loc.u_out), group=group)
    for rows, vals in parts:
        sd.u_out[rows, :] = vals
    return sd.u_out


def scene_prefers_exchanged_axes(sd):
    """The library's rule (pf_engine.hip: pf__axis_exchange_pays) -- rooms whose boundary nodes run along file x rather than z."""
    import ctypes
    from . import engine
    fn = engine.lib().pf__axis_exchange_pays
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    s = sd.as_struct()
    return bool(fn(ctypes.byref(s), None))


def measured_wall_scale(sd, rank, world, device, group=None, **engine_kw):
    """The factor on the wall planes' weights for this scene (pf_slab_wall_scale), measured by rank 0 on its device and handed to
    every rank, so that all cut the chain alike.  1.0 when the library declines (small scenes, fewer than 63 steps)."""
    from . import engine
    k = [1.0]
    together = world > 1 and dist.is_available() and dist.is_initialized()
    if rank == 0 or not together:  # (not together: a rank's cost model run alone, bench.py --emulate-via torch)
        kw = {f: engine_kw[f] for f in ("numerics", "air_variant", "air_chunk", "debug") if f in engine_kw}
        dev = device if isinstance(device, int) else (getattr(device, "index", None) or 0)
        k = [engine.slab_wall_scale(sd, world, dev, **kw) or 1.0]
    if together:
        dist.broadcast_object_list(k, src=0, group=group)
    return float(k[0])


def make_hip_runner(sd, rank, world, device, group=None, balance=True, along_z=None, wall_scale=1.0, **engine_kw):
    """wall_scale: factor on the wall planes' weights of the balanced cut; None = measured on the scene by rank 0 (measured_wall_scale;
    opt-in like the C chain's PF_MULTI_MEASURE_WEIGHTS: the measurement's noise is larger than the compiled-in weights' error).
    along_z: cut the chain along FILE Z (slab engines store the x and z axes exchanged; what rooms gain 11-15 % from, DESIGN.md
    5); None = the library's rule for the scene, like the C chain object (csrc/pf_multi.hip)."""
    import os
    if along_z is None:
        pairs_forced = engine_kw.get("pairs") or (engine_kw.get("air_variant", 0) & 255) in (40, 41)
        along_z = (world > 1 and not pairs_forced and not engine_kw.get("energy") and (sd.Nz - 2) // world >= 16
                   and not (int(engine_kw.get("debug", 0)) & 0x2000) and engine_kw.get("layout", 0) != 2 and scene_prefers_exchanged_axes(sd))
    if wall_scale is None:
        wall_scale = measured_wall_scale(sd, rank, world, device, group, **engine_kw) if (balance and world > 1 and not along_z) else 1.0
    loc, info = slab_mod.split(sd, world, rank, balance=balance, along_z=bool(along_z) and world > 1, wall_scale=wall_scale)
    info.wall_scale = wall_scale
    # temporally blocked step pairs in slab engines (four state grids per rank): measured on MI355X (1024^3, per-rank cost
    # model with an RCCL self-exchange, old / new on the same box) +3..14 % at 2 ranks, +5..10 % at 4, +7 % on the interior
    # slabs of 8 ranks (136 planes) and +0..2 % on its end slabs.  Default: on for slabs of at least 96 planes;
    # PFFDTD_SLAB_PAIRS=1 / 0 forces it on / off.  (The engine itself declines when the scene has no boundary-free box
    # or the y-z cross-section is small: pf_engine_set_spares returns 1 and it keeps stepping singly.)
    env = os.environ.get("PFFDTD_SLAB_PAIRS", "")
    engine_kw.setdefault("pairs", not getattr(info, "along_z", False) and (env == "1" or (env != "0" and loc.Nx - 2 >= 96)))
    st = HipSlabStepper(loc, info, device, **engine_kw)
    runner = SlabRunner(st, info, group)
    # one rank per GPU over RCCL: the planes by the library's own ncclSend / ncclRecv on the edge stream (PFFDTD_TORCH_P2P=1: torch's p2p)
    if (world > 1 and dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"
            and os.environ.get("PFFDTD_TORCH_P2P", "") != "1"):
        dev = device if isinstance(device, int) else (getattr(device, "index", None) or 0)
        runner.enable_native_rccl(dev)
    return runner, loc, info
