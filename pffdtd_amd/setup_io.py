"""Setup-side writers of the file contract: the pieces of the reference's `sim_setup()` that are pure
index/signal math, so an input folder can be produced (or GPU-prepared) without the reference's Python:

  SimConsts   python/fdtd/sim_consts.py:19-104      -> sim_consts.h5
  CartGrid    python/voxelizer/cart_grid.py:19-73    -> cart_grid.h5 (grid vectors; the voxelizer itself is out of scope)
  SimComms    python/fdtd/sim_comms.py:25-250        -> comms_out.h5 (trilinear source/receiver nodes, input signals,
                                                        bilinear differentiator)
  SimMats     python/fdtd/sim_mats.py:22-66          -> sim_mats.h5
  prep_folder python/fdtd/rotate_sim_data.py         -> rotate / fold / sort an existing folder for the multi-GPU engine

Same class and method names as the reference; storage through pffdtd_amd.h5io instead of h5py.
"""
import argparse
from pathlib import Path

import numpy as np

from . import h5io, synth


class SimConsts:
    def __init__(self, Tc, rh, h=None, SR=None, fmax=None, PPW=None, fcc=False):
        assert -20 <= Tc <= 50 and 10 <= rh <= 100
        c = 343.2 * np.sqrt(Tc / 20)
        assert (h is not None) or (SR is not None) or (fmax is not None and PPW is not None)
        l2 = 1.0 if fcc else 1 / 3
        l = np.sqrt(l2)
        l *= 0.999  # back off to remove the Nyquist mode
        l2 = l * l
        if h is not None:
            Ts = h / c * l
            SR = 1 / Ts
        elif SR is not None:
            Ts = 1 / SR
            h = c * Ts / l
        else:
            h = c / (fmax * PPW)
            Ts = h / c * l
            SR = 1 / Ts
        self.h, self.c, self.Ts, self.SR, self.l, self.l2, self.fcc, self.Tc, self.rh = h, c, Ts, SR, l, l2, fcc, Tc, rh

    def save(self, save_folder):
        p = Path(save_folder)
        p.mkdir(parents=True, exist_ok=True)
        f = p / "sim_consts.h5"
        first = True
        for k in ("c", "h", "Ts", "SR", "l", "l2"):
            h5io.write(f, k, np.float64(getattr(self, k)), append=not first)
            first = False
        h5io.write(f, "fcc_flag", np.int8(self.fcc))
        h5io.write(f, "Tc", np.float64(self.Tc))
        h5io.write(f, "rh", np.float64(self.rh))


class CartGrid:
    def __init__(self, h=None, offset=None, bmin=None, bmax=None, fcc=False):
        assert h is not None and offset is not None and bmin is not None and bmax is not None
        assert offset > 2.0  # three-layer halo for the ABCs (cart_grid.py:27-28)
        bmin, bmax = np.asarray(bmin, dtype=np.float64), np.asarray(bmax, dtype=np.float64)
        xyzmin0 = bmin - offset * h
        xyzmax0 = bmax + offset * h
        Nx, Ny, Nz = (np.int_(np.ceil((xyzmax0 - xyzmin0) / h)) + 1).tolist()
        if fcc:  # all dims even, so any can be rotated / halved
            Nx, Ny, Nz = Nx + (Nx % 2), Ny + (Ny % 2), Nz + (Nz % 2)
        self.xv = np.arange(Nx, dtype=np.float64) * h + xyzmin0[0]
        self.yv = np.arange(Ny, dtype=np.float64) * h + xyzmin0[1]
        self.zv = np.arange(Nz, dtype=np.float64) * h + xyzmin0[2]
        self.h, self.offset, self.Nx, self.Ny, self.Nz = h, offset, Nx, Ny, Nz
        self.Nxyz = np.array([Nx, Ny, Nz])
        self.Npts = int(np.prod(self.Nxyz))

    def save(self, save_folder):
        f = Path(save_folder) / "cart_grid.h5"
        Path(save_folder).mkdir(parents=True, exist_ok=True)
        h5io.write(f, "h", np.float64(self.h), append=False)
        for k in ("xv", "yv", "zv"):
            h5io.write(f, k, getattr(self, k))
        for k in ("Nx", "Ny", "Nz"):
            h5io.write(f, k, np.int64(getattr(self, k)))


class SimComms:
    def __init__(self, save_folder=None, h=None, Ts=None, l2=None, fcc_flag=0, xv=None, yv=None, zv=None):
        """Reads sim_consts.h5 / cart_grid.h5 from save_folder like the reference, or takes the values directly."""
        if save_folder is not None and h is None:
            p = Path(save_folder)
            c = p / "sim_consts.h5"
            h, Ts, l2, fcc_flag = (h5io.read(c, k) for k in ("h", "Ts", "l2", "fcc_flag"))
            g = p / "cart_grid.h5"
            xv, yv, zv = (h5io.read(g, k) for k in ("xv", "yv", "zv"))
        self.h, self.Ts, self.l2, self.fcc_flag = float(h), float(Ts), float(l2), int(fcc_flag)
        self.xv, self.yv, self.zv = (np.asarray(a, dtype=np.float64) for a in (xv, yv, zv))
        self.fcc = self.fcc_flag > 0
        if self.fcc:
            assert self.xv.size % 2 == 0 and self.yv.size % 2 == 0 and self.zv.size % 2 == 0
        self.save_folder = Path(save_folder) if save_folder is not None else None
        self._diff = False

    def get_linear_interp_weights(self, pos_xyz):
        """8 trilinear corner nodes and weights of a position (sim_comms.py:176-231); on the FCC subgrid the cube has
        twice the grid spacing and even-parity corners."""
        pos_xyz = np.asarray(pos_xyz, dtype=np.float64)
        vecs = [self.xv, self.yv, self.zv]
        Ny, Nz = self.yv.size, self.zv.size
        idx = np.empty(3, dtype=np.int64)
        alpha = np.zeros(3)
        for j in range(3):
            idx[j] = np.flatnonzero(vecs[j] >= pos_xyz[j])[0]
            alpha[j] = (vecs[j][idx[j]] - pos_xyz[j]) / self.h
        off = np.array([[0, 0, 0], [-1, 0, 0], [0, -1, 0], [0, 0, -1], [-1, -1, 0], [-1, 0, -1], [0, -1, -1], [-1, -1, -1]])
        if self.fcc:
            off = off * 2
            if np.mod(np.sum(idx), 2) == 1:
                idx[np.argmin(alpha)] += 1
            for j in range(3):
                alpha[j] = (vecs[j][idx[j]] - pos_xyz[j]) / (2 * self.h)
        alpha8 = np.ones(8)
        for i in range(8):
            for j in range(3):
                alpha8[i] *= (1 - alpha[j]) if off[i, j] == 0 else alpha[j]
        assert np.allclose(np.sum(alpha8), 1)
        corners = idx + off
        ixyz8 = corners @ np.array([Nz * Ny, Nz, 1])
        if self.fcc:
            assert np.all(np.mod(np.sum(corners, axis=-1), 2) == 0)
        return alpha8, ixyz8.astype(np.int64)

    def prepare_source_pts(self, Sxyz):
        self.in_alpha, self.in_ixyz = self.get_linear_interp_weights(Sxyz)

    def prepare_receiver_pts(self, Rxyz):
        Rxyz = np.atleast_2d(Rxyz)
        self.out_alpha = np.zeros((Rxyz.shape[0], 8))
        self.out_ixyz = np.zeros((Rxyz.shape[0], 8), dtype=np.int64)
        for r in range(Rxyz.shape[0]):
            self.out_alpha[r], self.out_ixyz[r] = self.get_linear_interp_weights(Rxyz[r])

    def prepare_source_signals(self, duration, sig_type="impulse"):
        Ts = self.Ts
        Nt = int(np.ceil(duration / Ts))
        sig = np.zeros((Nt,))
        if sig_type == "impulse":
            sig[0] = 1.0
        elif sig_type in ("hann10", "hann20"):
            N = int(sig_type[4:])
            n = np.arange(N)
            sig[:N] = 0.5 * (1.0 - np.cos(2 * np.pi * n / N))
        elif sig_type == "dhann30":
            N = 30
            n = np.arange(N)
            sig[:N] = np.cos(np.pi * n / N) * np.sin(np.pi * n / N)
        elif sig_type == "hann5ms":
            N = int(np.ceil(5e-3 / Ts))
            n = np.arange(N)
            sig[:N] = 0.5 * (1.0 - np.cos(2 * np.pi * n / N))
        # (an unknown sig_type silently gives a zero input, like the reference: SURVEY 4.1 quirk 9)
        self.in_sigs = self.in_alpha[:, None] * sig[None, :]
        self.in_sigs *= (0.5 * self.l2 / self.h) if self.fcc else (self.l2 / self.h)  # c^2 Ts^2 / cell volume

    def diff_source(self):
        """Bilinear-transform differentiator b = 2/Ts [1,-1], a = [1,1] (sim_comms.py:106-119)."""
        if self._diff:
            return
        x = self.in_sigs
        y = np.zeros_like(x)
        px = np.zeros(x.shape[0])
        py = np.zeros(x.shape[0])
        b0 = 2.0 / self.Ts
        for n in range(x.shape[1]):  # direct form, same recurrence as scipy.signal.lfilter(b, a, x)
            y[:, n] = b0 * x[:, n] - b0 * px - py
            px, py = x[:, n], y[:, n]
        self.in_sigs = y
        self._diff = True

    def check_for_clashes(self, bn_ixyz):
        for ix in (self.in_ixyz, self.out_ixyz):
            u = np.unique(ix)
            assert np.union1d(u, bn_ixyz).size == u.size + np.asarray(bn_ixyz).size, "source/receiver on a boundary node"

    def save(self, save_folder=None, compress=None):
        folder = Path(save_folder) if save_folder is not None else self.save_folder
        folder.mkdir(parents=True, exist_ok=True)
        f = folder / "comms_out.h5"
        out_ixyz = np.asarray(self.out_ixyz).reshape(-1)
        gz = int(compress) if compress else 0
        h5io.write(f, "in_ixyz", self.in_ixyz, append=False, gzip=gz)
        h5io.write(f, "out_ixyz", out_ixyz, gzip=gz)
        h5io.write(f, "out_alpha", self.out_alpha, gzip=gz)
        h5io.write(f, "out_reorder", np.arange(out_ixyz.size, dtype=np.int64), gzip=gz)
        h5io.write(f, "in_sigs", self.in_sigs, gzip=gz)
        h5io.write(f, "Ns", np.int64(self.in_ixyz.size))
        h5io.write(f, "Nr", np.int64(out_ixyz.size))
        h5io.write(f, "Nt", np.int64(self.in_sigs.shape[-1]))
        h5io.write(f, "diff", np.int8(self._diff))
        self.out_ixyz = out_ixyz


class SimMats:
    def __init__(self, save_folder):
        self.save_folder = Path(save_folder)

    def package(self, mat_files_dict, mat_list, read_folder):
        mat_list = sorted(m for m in mat_list if m != "_RIGID")
        assert mat_list == sorted(mat_files_dict.keys())  # sim_mats.py:30-36
        DEF_list = [np.asarray(h5io.read(Path(read_folder) / mat_files_dict[m], "DEF"), dtype=np.float64) for m in mat_list]
        self.save_folder.mkdir(parents=True, exist_ok=True)
        f = self.save_folder / "sim_mats.h5"
        h5io.write(f, "Nmat", np.int8(len(DEF_list)), append=False)
        Mb = np.zeros((len(DEF_list),), dtype=np.int8)
        for i, DEF in enumerate(DEF_list):
            assert DEF.ndim == 2 and DEF.shape[1] == 3
            h5io.write(f, f"mat_{i:02d}_DEF", DEF)
            Mb[i] = DEF.shape[0]
        h5io.write(f, "Mb", Mb)


def prep_folder(data_dir, rotate=True, fold=True, sort=True, out_dir=None, compress=0):
    """rotate_sim_data + fold_fcc_sim_data + sort_sim_data (sim_setup.py:127-133) on a folder, in place or into out_dir."""
    sim = synth.read_folder(data_dir)
    if rotate:
        synth.rotate_sim(sim)
    if fold and int(sim["sim_consts"]["fcc_flag"]) == 1:
        synth.fold_fcc(sim)
    if sort:
        synth.sort_sim(sim)
    synth.write_folder(sim, out_dir if out_dir is not None else data_dir, gzip=compress)
    return sim


def main():
    p = argparse.ArgumentParser(description="GPU-prepare a sim_data folder (rotate so Nx>=Ny>=Nz, fold the FCC subgrid, sort)")
    p.add_argument("--data_dir", required=True)
    p.add_argument("--out_dir", default=None)
    p.add_argument("--no-rotate", action="store_true")
    p.add_argument("--no-fold", action="store_true")
    p.add_argument("--no-sort", action="store_true")
    a = p.parse_args()
    prep_folder(a.data_dir, not a.no_rotate, not a.no_fold, not a.no_sort, a.out_dir)


if __name__ == "__main__":
    main()
