"""Z-slab decomposition of a SimData along the slowest axis (file Nx), one slab per GPU.

Re-thinks `split_data` + the per-GPU index localisation of the reference CUDA engine
(c_cuda/gpu_engine.h:516-662, 739-823) for the one-process-per-GPU runs: the cut itself comes from the
library (one implementation, `partition` below); here the node lists are cut by plane with vectorised
searches instead of per-GPU counting loops, and nothing requires pre-sorted input (the engine sorts its
lists itself).
"""
import copy

import numpy as np


def partition(sd, G, balance=False, along_z=False, wall_scale=1.0):
    """Owned plane ranges [(x0, x1)] per rank -- computed by the library (`pf_slab_partition_axis`, csrc/pf_multi.hip: partition; host code,
    no device needed): ONE implementation of the cut for the C chain and for the processes under torch.distributed.
    balance=False: the reference's rule, Nx//G planes each, +1 for the first Nx%G ranks (gpu_engine.h:532-550).  balance=True: equal
    estimated cost -- the end slabs of a room carry whole wall planes of boundary nodes, which the even split leaves unbalanced --,
    cuts kept clear of the sources.  Deterministic (every rank computes the same cut).  along_z: ranges of FILE Z instead of x (rooms,
    see split).  wall_scale: factor on the two wall-plane weights, as measured on the scene by the library (pf_slab_wall_scale)."""
    from . import engine
    nplanes = sd.Nz if along_z else sd.Nx
    if G < 1 or G >= nplanes:
        raise ValueError(f"need 1 <= ngpus < Nx (got {G}, Nx={nplanes})")  # gpu_engine.h:682
    return engine.slab_partition(sd, G, even=not balance, wall_scale=wall_scale, along_z=along_z)


def partition_weighted(sd, G, along_z=False, wall_scale=1.0):
    """The cost-balanced cut (partition(..., balance=True))."""
    return partition(sd, G, True, along_z, wall_scale)


class SlabInfo:
    def __init__(self, rank, G, x0, x1, Nx):
        self.rank, self.G, self.x0, self.x1 = rank, G, x0, x1
        self.first, self.last = rank == 0, rank == G - 1
        self.xlo = x0 - (0 if self.first else 1)      # global plane held in local plane 0
        self.xhi = x1 + (0 if self.last else 1)       # one past the last local plane
        self.Nxh = self.xhi - self.xlo                # local planes incl. ghosts (gpu_engine.h:755-760)
        # global planes this slab updates: its owned planes minus the global ghost planes 0 / Nx-1
        self.upd0 = max(x0, 1)
        self.upd1 = min(x1, Nx - 1)
        self.along_z = False                          # (split sets it: the plane numbers above are FILE Z then)


def split(sd, G, rank, balance=False, along_z=False, wall_scale=1.0):
    """Local SimData of slab `rank` of `G` (a shallow variant of `sd` with re-based lists) and its SlabInfo.
    balance=False: the reference's even split; True: cost-balanced cut (partition_weighted).
    along_z: the chain is cut along FILE Z instead of x -- for rooms whose engines store the grid with the x and z axes
    exchanged (pf_engine_layout; csrc/pf_multi.hip does the same): the slab then holds the file's columns z in [xlo, xhi) of
    every row, its local file has Nz = xhi - xlo, and `info`'s plane numbers are z."""
    nplanes = sd.Nz if along_z else sd.Nx
    parts = partition(sd, G, balance, along_z, wall_scale)
    x0, x1 = parts[rank]
    info = SlabInfo(rank, G, x0, x1, nplanes)
    info.along_z = along_z
    if info.upd1 - info.upd0 < 1:
        raise ValueError("a slab must own at least one interior plane")
    NzNy = sd.Ny * sd.Nz
    if along_z:
        nzl = info.Nxh

        def cut(idx):
            z = idx % sd.Nz
            return (z >= info.upd0) & (z < info.upd1)

        def local(idx):
            return (idx // sd.Nz) * nzl + (idx % sd.Nz - info.xlo)

        def owned(idx):
            z = idx % sd.Nz
            return (z >= x0) & (z < x1)
    else:
        off = info.xlo * NzNy                          # local index = global - off (gpu_engine.h:784-823)
        lo, hi = info.upd0 * NzNy, info.upd1 * NzNy

        def cut(idx):
            return (idx >= lo) & (idx < hi)

        def local(idx):
            return idx - off

        def owned(idx):
            return (idx >= x0 * NzNy) & (idx < x1 * NzNy)

    loc = copy.copy(sd)
    loc._keep = []
    if along_z:
        loc.Nz = info.Nxh
        loc.Npts = sd.Nx * sd.Ny * info.Nxh
    else:
        loc.Nx = info.Nxh
        loc.Npts = info.Nxh * NzNy
    kb = cut(sd.bn_ixyz)
    loc.bn_ixyz = np.ascontiguousarray(local(sd.bn_ixyz[kb]))
    loc.adj_bn = np.ascontiguousarray(sd.adj_bn[kb])
    loc.K_bn = np.ascontiguousarray(sd.K_bn[kb])
    loc.Nb = int(kb.sum())
    kl = cut(sd.bnl_ixyz)
    loc.bnl_ixyz = np.ascontiguousarray(local(sd.bnl_ixyz[kl]))
    loc.mat_bnl = np.ascontiguousarray(sd.mat_bnl[kl])
    loc.ssaf_bnl = np.ascontiguousarray(sd.ssaf_bnl[kl])
    if hasattr(sd, "saf_bnl"):
        loc.saf_bnl = sd.saf_bnl[kl]
    loc.Nbl = int(kl.sum())
    ka = cut(sd.bna_ixyz)
    loc.bna_ixyz = np.ascontiguousarray(local(sd.bna_ixyz[ka]))
    loc.Q_bna = np.ascontiguousarray(sd.Q_bna[ka])
    loc.Nba = int(ka.sum())
    ki = cut(sd.in_ixyz)
    loc.in_ixyz = np.ascontiguousarray(local(sd.in_ixyz[ki]))
    loc.in_sigs = np.ascontiguousarray(sd.in_sigs[ki])
    loc.Ns = int(ki.sum())
    # receivers read u1 at any owned plane (incl. a global ghost plane if someone asks for it)
    ko = owned(sd.out_ixyz)
    loc.out_ixyz = np.ascontiguousarray(local(sd.out_ixyz[ko]))
    loc.Nr = int(ko.sum())
    loc.out_rows = np.flatnonzero(ko)                  # rows of the global u_out this slab fills
    loc.out_reorder = np.arange(loc.Nr, dtype=np.int64)
    loc.u_out = np.zeros((loc.Nr, sd.Nt), dtype=np.float64)
    if sd.bn_mask is not None:                         # per-slab mask from own boundary nodes only (gpu_engine.h:791)
        flags = np.zeros((((loc.Npts - 1) // 8 + 1) * 8,), dtype=np.uint8)
        flags[loc.bn_ixyz] = 1
        loc.bn_mask = np.packbits(flags, bitorder="little")
    return loc, info


def merge_outputs(sd, locs):
    """Scatter the slabs' receiver rows back into the global u_out (gpu_engine.h:1066-1075)."""
    for loc in locs:
        sd.u_out[loc.out_rows, :] = loc.u_out
    return sd.u_out
