"""Air-absorption post-filters for simulated room impulse responses.

Mirrors of the reference's `air_abs` package: `get_air_absorption` (ISO 9613-1 coefficients,
python/air_abs/get_air_absorption.py:20-118), `apply_visco_filter` (Green's-function filter for Stokes' equation,
visco_filter.py:30-66), `apply_modal_filter` (modal / DCT domain, modal_filter.py:32-85) and `apply_ola_filter`
(overlap-add STFT, ola_filter.py:29-79).  Same arguments and return shapes; the loops of the reference are kept
sample by sample (they define the summation order), only the numba decorations are gone.  Host-side numpy / scipy:
these run once per RIR set after the simulation and have no GPU counterpart in the reference either.
"""
import numpy as np
from numpy import cos, exp, log, log10, pi, sqrt
from scipy.fft import dct, idct, irfft, rfft


def _iceil(x):
    return np.int_(np.ceil(x))


def get_air_absorption(freq_vec, temperature_celsius, rel_humidity_pnct, pressure_atmospheric_kPa=101.325):
    if not (pressure_atmospheric_kPa <= 200 and -20 <= temperature_celsius <= 50 and 10 <= rel_humidity_pnct <= 100):
        raise ValueError("air absorption: temperature -20..50 C, humidity 10..100 %, pressure <= 200 kPa")
    f, T, rh = freq_vec, temperature_celsius, rel_humidity_pnct
    f2, pi2 = f * f, pi * pi
    Tk, T01, T0 = T + 273.15, 273.16, 293.15
    pa = pr = 101.325
    thO, thN, XO, XN = 2239.1, 3352.0, 0.209, 0.781
    const = 2 * pi / 35 * (10 * log10(exp(2)))
    almO = const * XO * (thO / Tk) ** 2 * exp(-thO / Tk)
    almN = const * XN * (thN / Tk) ** 2 * exp(-thN / Tk)
    p, Tr = pa / pr, Tk / T0
    c = 343.2 * sqrt(Tr)
    c2 = c * c
    C = -6.8346 * (T01 / Tk) ** 1.261 + 4.6151
    h = rh * (10 ** C) * p
    frO = p * (24 + 4.04e4 * h * (0.02 + h) / (0.391 + h))
    frN = p * Tr ** (-0.5) * (9 + 280 * h * exp(-4.17 * (Tr ** (-1 / 3) - 1)))
    absfull1 = 8.686 * f2 * (1.84e-11 * sqrt(Tr) / p + Tr ** -2.5 * (0.01275 * (exp(-2239.1 / Tk) / (frO + f2 / frO))
                                                                      + 0.1068 * (exp(-3352.0 / Tk) / (frN + f2 / frN))))
    absClRo = 1.6e-10 * sqrt(Tr) * f2 / p
    eta = log(10) * 1.6e-11 / (4 * pi2) * (c2) * sqrt(Tr) / p
    absVibO = almO * (f / c) * (2 * (f / frO) / (1 + (f / frO) ** 2))
    absVibN = almN * (f / c) * (2 * (f / frN) / (1 + (f / frN) ** 2))
    absfull2 = absClRo + absVibO + absVibN
    assert np.allclose(absfull1, absfull2, rtol=1e-2)  # the two forms of the standard agree (get_air_absorption.py:96)
    etaO = almO * (c / pi2 / frO) * log(10) / 20
    return {"gamma_p": etaO / c, "gamma": eta / c, "etaO": etaO, "eta": eta, "almN": almN, "almO": almO, "c": c,
            "frO": frO, "frN": frN, "absVibN_dB": absVibN, "absVibO_dB": absVibO, "absClRo_dB": absClRo,
            "absfull_dB": absfull2, "absVibN_Np": absVibN * log(10) / 20, "absVibO_Np": absVibO * log(10) / 20,
            "absClRo_Np": absClRo * log(10) / 20, "absfull_Np": absfull2 * log(10) / 20}


def apply_visco_filter(x, Fs, Tc, rh, NdB=120, t_start=None):
    """Each input sample n is spread into a Gaussian of variance ~ n (the Green's function of Stokes' equation)."""
    rd = get_air_absorption(1, Tc, rh)
    g = rd["gamma_p"]
    Ts = 1 / Fs
    if t_start is None:
        t_start = Ts ** 2 / (2 * pi * g)
    x = np.atleast_2d(x)
    Nt0 = x.shape[-1]
    dt_end = Fs * sqrt(0.1 * log(10) * NdB * (Nt0 - 1) * Ts * g)
    Nt = Nt0 + _iceil(dt_end)
    y = np.zeros((x.shape[0], Nt))
    n_start = _iceil(t_start * Fs)
    assert n_start > 0
    y[:, :n_start] = x[:, :n_start]
    Tsg2, Tsg2pi = 2 * Ts * g, 2 * Ts * g * pi
    dt_fac = 0.1 * log(10) * NdB * (g * Ts)
    for n in range(n_start, Nt0):
        dt_int = _iceil(sqrt(dt_fac * n) / Ts)
        nv = np.arange(n - dt_int, n + dt_int + 1)
        assert n >= dt_int
        y[:, nv] += (Ts / sqrt(n * Tsg2pi)) * x[:, n][:, None] * exp(-((n - nv) * Ts) ** 2 / (n * Tsg2))[None, :]
    return np.squeeze(y)


def apply_modal_filter(x, Fs, Tc, rh, pad_t=0.0):
    """Time-reversed input drives one damped two-step recursion per DCT mode; the final state is the filtered signal."""
    Ts = 1 / Fs
    x = np.atleast_2d(x)
    Nt0 = x.shape[-1]
    Nt = _iceil(pad_t / Ts) + Nt0
    xp = np.zeros((x.shape[0], Nt))
    xp[:, :Nt0] = x
    wqTs = pi * (np.arange(Nt) / Nt)
    rd = get_air_absorption(wqTs / Ts / 2 / pi, Tc, rh)
    alphaq, c = rd["absfull_Np"], rd["c"]
    P0, P1 = np.zeros(xp.shape), np.zeros(xp.shape)
    fx = np.zeros(xp.shape)
    fx[:, 0] = 1
    Fm = dct(fx, type=2, norm="ortho", axis=-1)
    sigqTs = c * alphaq * Ts
    a1 = 2 * exp(-sigqTs) * cos(wqTs)
    a2 = -exp(-2 * sigqTs)
    Fmsig1 = Fm * (1 + sigqTs / 2) / (1 + sigqTs)
    Fmsig2 = Fm * (1 - sigqTs / 2) / (1 + sigqTs)
    u = np.zeros((xp.shape[0], Nt + 1))
    u[:, 1:] = xp[:, ::-1]
    for n in range(Nt):
        P0[:] = a1 * P1 + a2 * P0 + Fmsig1 * u[:, n + 1][:, None] - Fmsig2 * u[:, n][:, None]
        if n < Nt - 1:
            P1, P0 = P0, P1
    return np.squeeze(idct(P0, type=2, norm="ortho", axis=-1))


def apply_ola_filter(x, Fs, Tc, rh, Nw=1024):
    """Hann-windowed frames (75 % overlap), each attenuated by exp(-alpha(f) * c * t_frame) in the frequency domain."""
    Ts = 1 / Fs
    x = np.atleast_2d(x)
    Nt0 = x.shape[-1]
    Ha = np.int_(np.round(Nw * (1 - 0.75)))
    Nfft = np.int_(2 ** np.ceil(np.log2(Nw)))
    NF = _iceil((Nt0 + Nw) / Ha)
    Np = (NF - 1) * Ha - Nt0
    assert Nw - Ha <= Np < Nw
    Nfft_h = np.int_(Nfft // 2 + 1)
    xp = np.zeros((x.shape[0], Nw + Nt0 + Np))
    xp[:, Nw:Nw + Nt0] = x
    y = np.zeros((x.shape[0], Nt0 + Np))
    wa = 0.5 * (1 - cos(2 * pi * np.arange(Nw) / Nw))
    ws = wa / (3 / 8 * Nw / Ha)  # scaled for constant overlap-add
    rd = get_air_absorption(np.arange(Nfft_h) / Nfft * Fs, Tc, rh)
    c, absNp = rd["c"], rd["absfull_Np"]
    for i in range(xp.shape[0]):
        yp = np.zeros((xp.shape[-1],))
        for m in range(NF):
            na0 = m * Ha
            dist = c * Ts * (na0 - Nw / 2)
            xf = xp[i, na0:na0 + Nw]
            if dist < 0:  # pre-padding: no gain
                yp[na0:na0 + Nw] += ws * xf
            else:
                yp[na0:na0 + Nw] += ws * irfft(rfft(wa * xf, Nfft) * exp(-absNp * dist), Nfft)[:Nw]
        y[i] = yp[Nw:]
    return np.squeeze(y)
