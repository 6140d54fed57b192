"""Wall-impedance fits: absorption data -> the [Mb,3] DEF branch tables of sim_mats.h5.

Mirror of the reference's `materials/adm_funcs.py` (conversions :25-70, writers :73-123, `compute_Rf_from_DEF` :224-229,
`fit_to_Sabs_oct_11` :247-322) and of `build_mats.py` (the Sabine octave-band tables of its CTK / Musikverein
materials, :21-60).  Each branch m is a series RLC circuit with specific impedance  jw D + E + F/(jw);  a material is
the parallel connection of its branches.  Plotting is out of scope.

    python -m pffdtd_amd.materials --write_folder DIR      # regenerate the reference's data/materials/*.h5
"""
import argparse
from pathlib import Path

import numpy as np
import scipy.optimize as scpo
from numpy import log10, pi, sqrt

from . import h5io


def convert_nabs_to_R(nabs):
    """Normal-incidence absorption -> reflection coefficient."""
    nabs = np.float64(nabs)
    if not 0 <= nabs <= 1:
        raise ValueError("absorption coefficient outside [0,1]")
    return np.sqrt(1.0 - nabs)


def convert_Yn_to_R(Yn):
    assert np.all(Yn > 0.0)
    return (1.0 - Yn) / (1.0 + Yn)


def convert_R_to_Yn(R):
    assert np.all(R < 1.0)
    return (1.0 - R) / (1.0 + R)


def convert_R_to_Zn(R):
    return 1.0 / convert_R_to_Yn(R)


def convert_Sabs_to_Yn(Sabs, max_iter=100):
    """Sabine absorption -> specific admittance: Newton inversion of Paris' formula (adm_funcs.py:50-70)."""
    if Sabs > 0.9512:  # not reachable with a locally reacting wall
        Sabs = 0.9512
    if Sabs == 0:
        return 0
    fg = lambda g: 8.0 * g * (1 + g / (1 + g) - 2 * g * np.log((g + 1) / g))  # noqa: E731
    fgd = lambda g: -8.0 * (-4 * g ** 2 - 6 * g + 4 * (1 + g) ** 2 * g * np.log((g + 1) / g) - 1) / (1 + g) ** 2  # noqa: E731
    x_old, x_new, err, niter = Sabs / 8.0, 0, np.inf, 0
    while niter < max_iter and err > 1e-6:
        x_new = x_old - (fg(x_old) - Sabs) / fgd(x_old)
        niter += 1
        err = np.abs(1 - x_new / x_old)
        x_old = x_new
    return x_new


def write_freq_dep_mat(DEF, filename):
    DEF = np.atleast_2d(np.asarray(DEF, dtype=np.float64))
    if DEF.shape[1] != 3 or np.any(~np.isfinite(DEF)) or np.any(DEF < 0) or not np.all(np.sum(DEF > 0, axis=-1)):
        raise ValueError("DEF rows must be finite, non-negative and not all zero (rigid = no material in the scene)")
    h5io.write(Path(filename), "DEF", DEF, append=False)


def write_freq_ind_mat_from_Zn(Zn, filename):
    if not np.isfinite(Zn) or Zn < 0:
        raise ValueError("specific impedance must be finite and non-negative")
    write_freq_dep_mat(np.array([0, Zn, 0], dtype=np.float64), filename)


def write_freq_ind_mat_from_Yn(Yn, filename):
    if not np.isfinite(Yn) or Yn <= 0:
        raise ValueError("specific admittance must be finite and positive")
    write_freq_ind_mat_from_Zn(1 / Yn, filename)


def read_mat_DEF(filename):
    return np.asarray(h5io.read(Path(filename), "DEF"), dtype=np.float64)


def compute_Rf_from_DEF(jw, D, E, F):
    """Reflection coefficient / admittance of the parallel branches over jw (adm_funcs.py:224-229)."""
    Zn_br = jw[:, None] * D[None, :] + E + F[None, :] / jw[:, None]
    Yn = np.sum(1.0 / Zn_br, axis=-1)
    return (1.0 - Yn) / (1.0 + Yn), Yn, Zn_br, (Zn_br - 1.0) / (Zn_br + 1.0)


def _to_DEF(Ynm, dw, w0):  # peak admittance, half-power bandwidth, resonance (rad/s)
    return 1.0 / Ynm / dw, 1.0 / Ynm, w0 ** 2 / Ynm / dw


def fit_to_Sabs_oct_11(Sabs, filename=None):
    """11 Sabine octave-band coefficients (16 Hz .. 16 kHz) -> 11 resonant branches; only the branch peak admittances
    are optimised (Nelder-Mead on the summed absorption error over 10 Hz .. 20 kHz), bandwidths and centres fixed."""
    Sabs = np.asarray(Sabs, dtype=np.float64)
    assert Sabs.size == 11
    fv = np.logspace(log10(10), log10(20e3), 1000)
    jw = 1j * fv * 2 * pi
    fcv = 1000 * (2.0 ** np.arange(-6, 5))
    ymv, dwv, w0v = np.zeros(11), np.zeros(11), np.zeros(11)
    Y_target = np.zeros(fv.shape)
    for j in range(11):
        fc = fcv[j]
        Ynm = convert_Sabs_to_Yn(Sabs[j])
        i1 = 0 if j == 0 else np.flatnonzero(fv >= fc / sqrt(2))[0]
        i2 = fv.size if j == 10 else np.flatnonzero(fv >= fc * sqrt(2))[0]
        Y_target[i1:i2] = Ynm
        w0 = 2 * pi * fc
        ymv[j], dwv[j], w0v[j] = Ynm, w0 / sqrt(2), w0
    R_target = (1.0 - Y_target) / (1.0 + Y_target)

    def cost(ym):
        x0 = np.c_[ym, dwv, w0v].flat[:]
        if np.any(x0 < 0):
            return np.finfo(np.float64).max
        x0 = x0.reshape((-1, 3))
        D, E, F = _to_DEF(x0[:, 0], x0[:, 1], x0[:, 2])
        Rf_opt = compute_Rf_from_DEF(jw, D, E, F)[0]
        return np.sum(np.abs((1 - np.abs(Rf_opt) ** 2) - (1 - np.abs(R_target) ** 2)))

    res = scpo.minimize(cost, ymv, method="Nelder-Mead")
    assert cost(res.x) <= cost(ymv)
    DEF = np.c_[_to_DEF(res.x, dwv, w0v)]
    if filename is not None:
        write_freq_dep_mat(DEF, filename)
    return DEF


# Sabine coefficients of the reference's example materials, 16 Hz .. 16 kHz octave bands (build_mats.py:24-47)
SABINE_OCT_11 = {
    "mv_chairs": [0.22, 0.22, 0.22, 0.22, 0.26, 0.3, 0.33, 0.34, 0.34, 0.34, 0.34],
    "mv_floor": [0.14, 0.14, 0.14, 0.14, 0.1, 0.06, 0.08, 0.1, 0.1, 0.1, 0.1],
    "mv_plasterboard": [0.15, 0.15, 0.15, 0.15, 0.1, 0.06, 0.04, 0.04, 0.05, 0.05, 0.05],
    "mv_window": [0.35, 0.35, 0.35, 0.35, 0.25, 0.18, 0.12, 0.07, 0.04, 0.04, 0.04],
    "mv_wood": [0.25, 0.25, 0.25, 0.25, 0.15, 0.1, 0.09, 0.08, 0.07, 0.07, 0.07],
    "ctk_acoustic_panel": [0.2, 0.2, 0.42, 0.89, 1, 1, 1, 1, 1, 1, 1],
    "ctk_altar": [0.25, 0.25, 0.25, 0.25, 0.15, 0.1, 0.09, 0.08, 0.07, 0.07, 0.07],
    "ctk_audience": [0.1, 0.1, 0.1, 0.1, 0.07, 0.08, 0.1, 0.1, 0.11, 0.11, 0.11],
    "ctk_carpet": [0.08, 0.08, 0.08, 0.08, 0.24, 0.57, 0.69, 0.71, 0.73, 0.73, 0.73],
    "ctk_ceiling": [0.19, 0.19, 0.19, 0.19, 0.06, 0.05, 0.08, 0.07, 0.05, 0.05, 0.05],
    "ctk_chair": [0.44, 0.44, 0.44, 0.44, 0.56, 0.67, 0.74, 0.83, 0.87, 0.87, 0.87],
    "ctk_tile": [0.015, 0.015, 0.015, 0.015, 0.015, 0.005, 0.005, 0.005, 0.005, 0.005, 0.005],
    "ctk_walls": [0.19, 0.19, 0.19, 0.19, 0.06, 0.05, 0.08, 0.07, 0.05, 0.05, 0.05],
    "ctk_window": [0.35, 0.35, 0.35, 0.35, 0.25, 0.18, 0.12, 0.07, 0.04, 0.04, 0.04],
}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--write_folder", required=True)
    a = p.parse_args()
    folder = Path(a.write_folder)
    folder.mkdir(parents=True, exist_ok=True)
    for name, sabs in SABINE_OCT_11.items():
        fit_to_Sabs_oct_11(np.array(sabs), folder / f"{name}.h5")
        print(f"--MATS: wrote {name}.h5")


if __name__ == "__main__":
    main()
