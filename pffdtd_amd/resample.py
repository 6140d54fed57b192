"""Band-limited sinc-interpolation resampler = resampy's `resample(x, sr_orig, sr_new, filter='kaiser_best')`, which the
reference's `ProcessOutputs.resample` calls on the filtered receiver signals (python/fdtd/process_outputs.py:154-166).

resampy is a third-party dependency (python/pip_requirements.txt:18 `resampy >= 0.2.2`, conda_pffdtd.yml:20), absent from
/root/reference and from this image: **parity unpinned**.  This is a restatement of its published algorithm (J. O. Smith,
"Digital Audio Resampling Home Page", the `resample_f` loop of resampy/interpn.py and `sinc_window` of resampy/filters.py)
with the published 'kaiser_best' design: 64 zero crossings, 2^9 table samples per crossing, roll-off 0.9475937167399596 of
Nyquist, Kaiser window beta 14.769656459379492 (resampy ships the table as a data file; it is recomputed here).  Output
time grid: t_k = k * sr_orig / sr_new (resampy >= 0.3; 0.2.x accumulates the increment instead, which differs by rounding).
tests/test_resample.py checks it against a literal per-sample transcription of the loop and against analytic signals.

Host-side DSP on Nr x Nt samples (out of the hot path): numpy, no device work.
"""
import numpy as np
from scipy.signal.windows import kaiser

KAISER_BEST = dict(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492)
_cache = {}


def sinc_window(num_zeros, precision, rolloff, beta):
    """Right half of the windowed sinc, sampled 2^precision times per zero crossing (resampy/filters.py:sinc_window)."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def kaiser_best():
    if "kb" not in _cache:
        _cache["kb"] = sinc_window(**KAISER_BEST)
    return _cache["kb"]


def resample(x, sr_orig, sr_new, axis=-1):
    """x resampled from sr_orig to sr_new along `axis`; int(n * sr_new / sr_orig) output samples."""
    if sr_orig <= 0 or sr_new <= 0:
        raise ValueError("sample rates must be positive")
    x = np.asarray(x)
    ratio = float(sr_new) / float(sr_orig)
    n_orig = x.shape[axis]
    n_out = int(n_orig * ratio)
    if n_out < 1:
        raise ValueError(f"input of {n_orig} samples is too short for the ratio {ratio}")
    win, num_table = kaiser_best()
    if ratio < 1:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    index_step = int(scale * num_table)
    nwin = win.shape[0]

    xm = np.moveaxis(x, axis, -1)
    dt = xm.dtype if np.issubdtype(xm.dtype, np.floating) else np.float64
    X = np.ascontiguousarray(xm, dtype=dt).reshape(-1, n_orig)
    Y = np.zeros((X.shape[0], n_out), dtype=dt)
    t = np.arange(n_out) * (1.0 / ratio)
    n = t.astype(np.int64)
    # the taps are accumulated in resampy's order (left wing outwards, then right wing outwards), one tap of every output
    # sample per pass
    for wing in (0, 1):
        frac = scale * (t - n)
        if wing:
            frac = scale - frac
        index_frac = frac * num_table
        offset = index_frac.astype(np.int64)
        eta = index_frac - offset
        lim = (n + 1) if wing == 0 else (n_orig - n - 1)
        count = np.minimum(lim, (nwin - offset) // index_step)
        for i in range(int(count.max(initial=0))):
            live = np.nonzero(count > i)[0]
            k = offset[live] + i * index_step
            w = win[k] + eta[live] * delta[k]
            src = (n[live] - i) if wing == 0 else (n[live] + i + 1)
            Y[:, live] += w * X[:, src]
    return np.moveaxis(Y.reshape(*xm.shape[:-1], n_out), -1, axis)
