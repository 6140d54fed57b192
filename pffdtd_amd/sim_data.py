"""Host-side `SimData`: the Python mirror of the reference's loader and its `struct SimData`.

Restates what `load_sim_data` derives from the four input files (c_cuda/fdtd_data.h:99-718), plus
`scale_input` / `rescale_output` / `write_outputs` / `print_last_samples` (fdtd_data.h:863-980), on
numpy arrays, and marshals the result into the C-ABI `pf_simdata` (include/pffdtd_hip.h).
Checks that are `assert`s in the reference raise `ValueError` here.
"""
import ctypes
from pathlib import Path

import numpy as np

from . import h5io, synth

MMb = 12  # fdtd_data.h:33
MNm = 64  # fdtd_data.h:35

_vp = ctypes.c_void_p


class PfSimData(ctypes.Structure):
    """ctypes image of `pf_simdata` (include/pffdtd_hip.h); field order = struct SimData, fdtd_data.h:38-76."""
    _fields_ = [
        ("bn_ixyz", _vp), ("bnl_ixyz", _vp), ("bna_ixyz", _vp), ("Q_bna", _vp), ("in_ixyz", _vp),
        ("out_ixyz", _vp), ("out_reorder", _vp), ("adj_bn", _vp), ("ssaf_bnl", _vp), ("bn_mask", _vp),
        ("mat_bnl", _vp), ("K_bn", _vp), ("in_sigs", _vp), ("u_out", _vp),
        ("Ns", ctypes.c_int64), ("Nr", ctypes.c_int64), ("Nt", ctypes.c_int64), ("Npts", ctypes.c_int64),
        ("Nx", ctypes.c_int64), ("Ny", ctypes.c_int64), ("Nz", ctypes.c_int64), ("Nb", ctypes.c_int64),
        ("Nbl", ctypes.c_int64), ("Nba", ctypes.c_int64),
        ("l", ctypes.c_double), ("l2", ctypes.c_double),
        ("fcc_flag", ctypes.c_int8), ("NN", ctypes.c_int8), ("Nm", ctypes.c_int8), ("Mb", _vp),
        ("mat_quads", _vp), ("mat_beta", _vp), ("infac", ctypes.c_double),
        ("sl2", ctypes.c_double), ("lo2", ctypes.c_double), ("a2", ctypes.c_double), ("a1", ctypes.c_double),
        ("real_bytes", ctypes.c_int32),
    ]


def _ptr(a):
    return _vp(a.ctypes.data) if a is not None else _vp(None)


def abc_nodes(Nx, Ny, Nz, fcc_flag):
    """ABC node list and Q (1 face, 2 edge, 3 corner): fdtd_data.h:621-675 (python: sim_fdtd.py:867-886).

    For the folded FCC grid the list is built on the unfolded extent Nyf = 2(Ny-1), mapped through the
    fold and sorted with its Q companion.
    """
    Nyf = 2 * (Ny - 1) if fcc_flag == 2 else Ny
    Nba = 2 * (Nx * Nyf + Nx * Nz + Nyf * Nz) - 12 * (Nx + Nyf + Nz) + 56
    if fcc_flag > 0:
        Nba //= 2
    iys = np.arange(1, Nyf - 1, dtype=np.int64)
    izs = np.arange(1, Nz - 1, dtype=np.int64)
    qy = ((iys == 1) | (iys == Nyf - 2)).astype(np.int8)
    qz = ((izs == 1) | (izs == Nz - 2)).astype(np.int8)
    Q2 = qy[:, None] + qz[None, :]
    par2 = (iys[:, None] + izs[None, :]) % 2
    iyf = np.where(iys >= Nyf // 2, Nyf - iys - 1, iys) if fcc_flag == 2 else iys
    off2 = iyf[:, None] * Nz + izs[None, :]
    # a plane's node pattern depends only on (is it an x-shell plane, parity of ix): build the <=4 patterns once;
    # row-major nonzero() order == the reference's iy,iz loop order
    pats = {}
    for qx in (0, 1):
        for par in ((0, 1) if fcc_flag > 0 else (None,)):
            sel = (Q2 + qx) > 0
            if par is not None:
                sel = sel & (par2 == par)  # (ix+iy+iz) even  <=>  (iy+iz)%2 == ix%2
            pats[(qx, par)] = (off2[sel], (Q2[sel] + qx).astype(np.int8))
    idx_parts, q_parts = [], []
    for ix in range(1, Nx - 1):
        key = (1 if ix in (1, Nx - 2) else 0, (ix % 2) if fcc_flag > 0 else None)
        o, q = pats[key]
        idx_parts.append(ix * Nz * Ny + o)
        q_parts.append(q)
    bna = np.concatenate(idx_parts).astype(np.int64)
    Q = np.concatenate(q_parts).astype(np.int8)
    if bna.size != Nba:
        raise ValueError(f"ABC node count {bna.size} != {Nba}")  # fdtd_data.h:653
    if fcc_flag == 2:
        # qsort_keys (helper_funcs.h:115-132) is not a stable sort, but a folded node appears at most once
        # (the two unfolded pre-images of a row differ in parity), so key order is unambiguous.
        o = np.argsort(bna, kind="stable")
        bna, Q = bna[o], Q[o]
    return bna, Q


class SimData:
    """Derived engine inputs for one precision. Build with from_sim()/from_folder()."""

    def __init__(self):
        self._keep = []

    # ---- load_sim_data, fdtd_data.h:99-718 ----
    @classmethod
    def from_sim(cls, sim, precision="double", build_mask=True):
        sd = cls()
        real = {"double": np.float64, "single": np.float32, 2: np.float64, 1: np.float32}[precision]
        sd.real = real
        sd.real_bytes = np.dtype(real).itemsize
        k, v, c, m = sim["sim_consts"], sim["vox_out"], sim["comms_out"], sim["sim_mats"]
        sd.l, sd.l2, sd.Ts = float(k["l"]), float(k["l2"]), float(k["Ts"])
        sd.fcc_flag = int(k["fcc_flag"])
        if not 0 <= sd.fcc_flag <= 2:
            raise ValueError("fcc_flag must be 0, 1 or 2")  # :165
        if sd.fcc_flag > 0:
            if not (sd.l2 <= 1.0 and sd.l <= 1.0):
                raise ValueError("CFL violated (FCC)")  # :175-176
            sd.NN = 12
        else:
            if not (sd.l2 <= 1.0 / 3.0 and sd.l <= np.sqrt(1.0 / 3.0)):
                raise ValueError("CFL violated (Cartesian)")  # :180-181
            sd.NN = 6
        # coefficients (:186-194); EPS from fdtd_common.h:55,68
        EPS = 0.0 if real is np.float64 else 1.19209289e-07
        lfac = 0.25 if sd.fcc_flag > 0 else 1.0
        dsl2 = (1.0 + EPS) * lfac * sd.l2
        da1 = 2.0 - dsl2 * sd.NN
        da2 = lfac * sd.l2
        sd.a1, sd.a2, sd.sl2, sd.lo2 = real(da1), real(da2), real(dsl2), real(0.5 * sd.l)

        sd.Nx, sd.Ny, sd.Nz, sd.Nb = int(v["Nx"]), int(v["Ny"]), int(v["Nz"]), int(v["Nb"])
        sd.Npts = sd.Nx * sd.Ny * sd.Nz
        bn_ixyz = np.ascontiguousarray(v["bn_ixyz"], dtype=np.int64)
        adj_bool = np.asarray(v["adj_bn"]).astype(np.bool_)
        mat_bn = np.ascontiguousarray(v["mat_bn"], dtype=np.int8)
        saf_bn = np.ascontiguousarray(v["saf_bn"], dtype=np.float64)
        if bn_ixyz.shape != (sd.Nb,) or adj_bool.shape != (sd.Nb, sd.NN) or mat_bn.shape != (sd.Nb,) \
                or saf_bn.shape != (sd.Nb,):
            raise ValueError("vox_out.h5 dataset shapes do not match Nb/NN")  # :243,256-257,274,281
        if sd.fcc_flag > 0:  # :283-289: Real-rounded constant times a double, then rounded to Real
            ssaf_bn = (np.float64(real(0.5 / np.sqrt(2.0))) * saf_bn).astype(real)
        else:
            ssaf_bn = saf_bn.astype(real)

        sd.Nt, sd.Ns, sd.Nr = int(c["Nt"]), int(c["Ns"]), int(c["Nr"])
        sd.diff = bool(c["diff"])
        sd.in_ixyz = np.ascontiguousarray(c["in_ixyz"], dtype=np.int64)
        sd.out_ixyz = np.ascontiguousarray(c["out_ixyz"], dtype=np.int64)
        sd.out_reorder = np.ascontiguousarray(c["out_reorder"], dtype=np.int64)
        sd.in_sigs = np.array(c["in_sigs"], dtype=np.float64, order="C")  # private copy: scale_input edits it
        if sd.in_ixyz.shape != (sd.Ns,) or sd.out_ixyz.shape != (sd.Nr,) or sd.out_reorder.shape != (sd.Nr,) \
                or sd.in_sigs.shape != (sd.Ns, sd.Nt):
            raise ValueError("comms_out.h5 dataset shapes do not match Ns/Nr/Nt")  # :342,355,359,372-373
        if sd.real_bytes == 4 and not sd.diff:
            raise ValueError("single precision requires a differentiated input (diff=1)")  # :392

        sd.Nm = int(m["Nmat"])
        if sd.Nm > MNm:
            raise ValueError("too many materials")  # :412
        sd.Mb = np.ascontiguousarray(m["Mb"], dtype=np.int8)
        quads = np.zeros((max(sd.Nm, 1) * MMb, 4), dtype=real)  # b, bd, bDh, bFh (:441-456)
        beta = np.zeros((max(sd.Nm, 1),), dtype=real)
        sd.DEF = []
        for i in range(sd.Nm):
            DEF = np.asarray(m[f"mat_{i:02d}_DEF"], dtype=np.float64)
            if DEF.shape != (int(sd.Mb[i]), 3) or sd.Mb[i] > MMb:
                raise ValueError(f"mat_{i:02d}_DEF shape {DEF.shape} does not match Mb")  # :430-432
            sd.DEF.append(DEF)
            acc = real(0.0)
            for j in range(int(sd.Mb[i])):
                D, E, F = (float(x) for x in DEF[j])
                Dh, Eh, Fh = D / sd.Ts, E, F * sd.Ts
                b = 1.0 / (2.0 * Dh + Eh + 0.5 * Fh)
                bd = b * (2.0 * Dh - Eh - 0.5 * Fh)
                if not (np.isfinite(b) and np.isfinite(bd)):
                    raise ValueError("non-finite material coefficient")  # :449-450
                quads[MMb * i + j] = (real(b), real(bd), real(b * Dh), real(b * Fh))
                acc = real(acc + real(b))  # accumulated in Real (:457)
            beta[i] = acc
        sd.mat_quads, sd.mat_beta = quads, beta

        # checks and repacking (:507-614)
        if sd.Nb:
            ix, iy, iz = synth._ind2sub(bn_ixyz, sd.Ny, sd.Nz)
            ok = (ix > 0) & (iy > 0) & (iz > 0) & (ix < sd.Nx - 1) & (iy < sd.Ny - 1) & (iz < sd.Nz - 1)
            if not ok.all():
                raise ValueError("boundary node outside the grid interior")  # :510, fdtd_common.h:83-101
            if adj_bool.all(axis=1).any():
                raise ValueError("boundary node with all neighbours adjacent")  # :524
            if (mat_bn[~adj_bool.any(axis=1)] != -1).any():
                raise ValueError("isolated node must be rigid (mat -1)")  # :525
        weights = (1 << np.arange(sd.NN, dtype=np.uint16)).astype(np.uint16)
        sd.adj_bn = (adj_bool.astype(np.uint16) * weights[None, :]).sum(axis=1).astype(np.uint16)  # :532-538
        sd.K_bn = adj_bool.sum(axis=1).astype(np.int8)  # :553-560
        sd.bn_ixyz = bn_ixyz
        if build_mask:  # :567-572 (the HIP engine builds its own padded mask and accepts NULL here)
            flags = np.zeros((((sd.Npts - 1) // 8 + 1) * 8,), dtype=np.uint8)
            flags[bn_ixyz] = 1
            sd.bn_mask = np.packbits(flags, bitorder="little")
        else:
            sd.bn_mask = None
        lossy = mat_bn >= 0  # :595-614, order preserved
        sd.Nbl = int(lossy.sum())
        sd.mat_bnl = np.ascontiguousarray(mat_bn[lossy])
        sd.ssaf_bnl = np.ascontiguousarray(ssaf_bn[lossy])
        sd.bnl_ixyz = np.ascontiguousarray(bn_ixyz[lossy])
        if sd.Nbl and (sd.mat_bnl >= sd.Nm).any():
            raise ValueError("material index out of range")
        sd.saf_bnl = saf_bn[lossy]  # unscaled, kept for the energy diagnostic (sim_fdtd.py:73)
        sd.bna_ixyz, sd.Q_bna = abc_nodes(sd.Nx, sd.Ny, sd.Nz, sd.fcc_flag)
        sd.Nba = int(sd.bna_ixyz.size)
        sd.u_out = np.zeros((sd.Nr, sd.Nt), dtype=np.float64)  # :678
        sd.infac = 1.0
        sd.h = float(k["h"]) if "h" in k else None
        sd.c = float(k["c"]) if "c" in k else None
        return sd

    @classmethod
    def from_folder(cls, data_dir, precision="double", build_mask=True):
        sd = cls.from_sim(synth.read_folder(data_dir), precision, build_mask=build_mask)
        sd.data_dir = Path(data_dir)
        return sd

    # ---- scale_input / rescale_output, fdtd_data.h:879-925 ----
    def scale_input(self):
        fi = np.finfo(self.real)
        max_in = float(np.max(np.abs(self.in_sigs))) if self.in_sigs.size else 0.0
        aexp = 0.5
        # REAL_MAX_EXP/REAL_MIN_EXP are <float.h>'s *_MAX_EXP/*_MIN_EXP = numpy's maxexp / minexp+1
        pow2 = int(np.round(aexp * fi.maxexp + (1 - aexp) * (fi.minexp + 1)))
        norm1 = 2.0 ** pow2
        inv_infac = norm1 / max_in
        self.infac = 1.0 / inv_infac
        self.in_sigs *= inv_infac
        return self.infac

    def rescale_output(self):
        self.u_out *= self.infac

    # ---- write_outputs / print_last_samples, fdtd_data.h:863-876,928-980 ----
    def write_outputs(self, data_dir=None):
        data_dir = Path(data_dir if data_dir is not None else getattr(self, "data_dir", "."))
        h5io.write(data_dir / "sim_outs.h5", "u_out", self.u_out[self.out_reorder, :], append=False)

    def print_last_samples(self, Np=5, file=None):
        print("RAW OUTPUTS", file=file)
        for nr in range(self.Nr):
            print(f"receiver {nr}", file=file)
            for n in range(max(self.Nt - Np, 0), self.Nt):
                print(f"sample {n}: {self.u_out[self.out_reorder[nr], n]:.16e}", file=file)

    # ---- C-ABI marshalling ----
    def as_struct(self):
        s = PfSimData()
        for name in ("bn_ixyz", "bnl_ixyz", "bna_ixyz", "Q_bna", "in_ixyz", "out_ixyz", "out_reorder", "adj_bn",
                     "ssaf_bnl", "bn_mask", "mat_bnl", "K_bn", "in_sigs", "u_out", "Mb", "mat_quads", "mat_beta"):
            a = getattr(self, name)
            if a is None:
                continue
            if not a.flags["C_CONTIGUOUS"]:
                raise ValueError(f"{name} must be contiguous")
            setattr(s, name, _ptr(a))
        for name in ("Ns", "Nr", "Nt", "Npts", "Nx", "Ny", "Nz", "Nb", "Nbl", "Nba", "l", "l2", "fcc_flag", "NN",
                     "Nm", "infac", "real_bytes"):
            setattr(s, name, getattr(self, name))
        for name in ("sl2", "lo2", "a2", "a1"):
            setattr(s, name, float(getattr(self, name)))
        self._keep.append(s)
        return s
