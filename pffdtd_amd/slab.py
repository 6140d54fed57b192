"""Z-slab decomposition of a SimData along the slowest axis (file Nx), one slab per GPU.

Re-thinks `split_data` + the per-GPU index localisation of the reference CUDA engine
(c_cuda/gpu_engine.h:516-662, 739-823): same partition rule (Nx/G planes each, remainder to the first
ranks, one ghost plane on every interior side), but the node lists are cut by plane with vectorised
searches instead of per-GPU counting loops, and nothing here requires pre-sorted input (the engine
sorts its lists itself).
"""
import copy

import numpy as np


def partition(Nx, G):
    """Owned plane ranges [x0, x1) per rank: Nx//G planes each, +1 for the first Nx%G ranks (gpu_engine.h:532-550)."""
    if G < 1 or G >= Nx:
        raise ValueError(f"need 1 <= ngpus < Nx (got {G}, Nx={Nx})")  # gpu_engine.h:682
    base, rem = divmod(Nx, G)
    sizes = [base + (1 if g < rem else 0) for g in range(G)]
    x0 = np.concatenate([[0], np.cumsum(sizes)])
    return [(int(x0[g]), int(x0[g + 1])) for g in range(G)]


# measured on MI355X (1024^2 planes): one full plane of lossy (Mb=11) boundary nodes costs about as much as 24 planes
# of interior update, a full plane of rigid boundary nodes about 5 (k_boundary vs k_air_cart_lean, profiles/r01_*)
LOSSY_PLANE_EQ = 24.0
RIGID_PLANE_EQ = 5.0


def partition_weighted(sd, G):
    """Owned plane ranges balanced by estimated cost instead of plane count: the end slabs of a room carry whole
    wall planes of boundary nodes, which the reference's even split (gpu_engine.h:532-550) leaves unbalanced.
    Deterministic (every rank computes the same cut)."""
    Nx = sd.Nx
    if G < 1 or G >= Nx:
        raise ValueError(f"need 1 <= ngpus < Nx (got {G}, Nx={Nx})")
    if G == 1:
        return [(0, Nx)]
    NzNy = sd.Ny * sd.Nz
    nb = np.bincount(sd.bn_ixyz // NzNy, minlength=Nx).astype(np.float64)
    nl = np.bincount(sd.bnl_ixyz // NzNy, minlength=Nx).astype(np.float64) if sd.Nbl else np.zeros(Nx)
    mb_scale = 1.0
    if sd.Nbl:
        mb_scale = float(np.mean(sd.Mb[sd.mat_bnl])) / 11.0
    cost = np.ones(Nx)
    cost[0] = cost[-1] = 0.0  # global ghost planes are not updated
    cost += (LOSSY_PLANE_EQ * mb_scale * nl + RIGID_PLANE_EQ * (nb - nl)) / NzNy
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = [0]
    for g in range(1, G):
        target = cum[-1] * g / G
        x = int(np.searchsorted(cum, target))
        x = max(x, cuts[-1] + 2)           # every slab updates at least one plane
        x = min(x, Nx - 2 * (G - g))
        cuts.append(x)
    cuts.append(Nx)
    return [(cuts[g], cuts[g + 1]) for g in range(G)]


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


def split(sd, G, rank, balance=False):
    """Local SimData of slab `rank` of `G` (a shallow variant of `sd` with re-based lists) and its SlabInfo.
    balance=False: the reference's even split; True: cost-balanced cut (partition_weighted)."""
    parts = partition_weighted(sd, G) if balance else partition(sd.Nx, G)
    x0, x1 = parts[rank]
    info = SlabInfo(rank, G, x0, x1, sd.Nx)
    if info.upd1 - info.upd0 < 1:
        raise ValueError("a slab must own at least one interior plane")
    NzNy = sd.Ny * sd.Nz
    off = info.xlo * NzNy                              # local index = global - off (gpu_engine.h:784-823)
    lo, hi = info.upd0 * NzNy, info.upd1 * NzNy

    def cut(idx):
        return (idx >= lo) & (idx < hi)

    loc = copy.copy(sd)
    loc._keep = []
    loc.Nx = info.Nxh
    loc.Npts = info.Nxh * NzNy
    kb = cut(sd.bn_ixyz)
    loc.bn_ixyz = np.ascontiguousarray(sd.bn_ixyz[kb] - off)
    loc.adj_bn = np.ascontiguousarray(sd.adj_bn[kb])
    loc.K_bn = np.ascontiguousarray(sd.K_bn[kb])
    loc.Nb = int(kb.sum())
    kl = cut(sd.bnl_ixyz)
    loc.bnl_ixyz = np.ascontiguousarray(sd.bnl_ixyz[kl] - off)
    loc.mat_bnl = np.ascontiguousarray(sd.mat_bnl[kl])
    loc.ssaf_bnl = np.ascontiguousarray(sd.ssaf_bnl[kl])
    if hasattr(sd, "saf_bnl"):
        loc.saf_bnl = sd.saf_bnl[kl]
    loc.Nbl = int(kl.sum())
    ka = cut(sd.bna_ixyz)
    loc.bna_ixyz = np.ascontiguousarray(sd.bna_ixyz[ka] - off)
    loc.Q_bna = np.ascontiguousarray(sd.Q_bna[ka])
    loc.Nba = int(ka.sum())
    ki = cut(sd.in_ixyz)
    loc.in_ixyz = np.ascontiguousarray(sd.in_ixyz[ki] - off)
    loc.in_sigs = np.ascontiguousarray(sd.in_sigs[ki])
    loc.Ns = int(ki.sum())
    # receivers read u1 at any owned plane (incl. a global ghost plane if someone asks for it)
    ko = (sd.out_ixyz >= x0 * NzNy) & (sd.out_ixyz < x1 * NzNy)
    loc.out_ixyz = np.ascontiguousarray(sd.out_ixyz[ko] - off)
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
