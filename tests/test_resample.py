"""pffdtd_amd.resample (resampy 'kaiser_best' restated; parity unpinned -- resampy is in neither /root/reference nor this
image) against a literal per-sample transcription of the published interpolation loop and against analytic signals."""
import numpy as np
import pytest

from pffdtd_amd import resample as rs


def loop_resample(x, sr_orig, sr_new):
    """One output sample at a time, one tap at a time: the loop of resampy/interpn.py as published."""
    ratio = float(sr_new) / sr_orig
    win, num_table = rs.sinc_window(**rs.KAISER_BEST)
    if ratio < 1:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    time_increment = 1.0 / ratio
    index_step = int(scale * num_table)
    n_orig, n_out, nwin = len(x), int(len(x) * ratio), len(win)
    y = np.zeros(n_out, dtype=x.dtype)
    for t in range(n_out):
        time_register = t * time_increment
        n = int(time_register)
        frac = scale * (time_register - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        for i in range(min(n + 1, (nwin - offset) // index_step)):
            y[t] += (win[offset + i * index_step] + eta * delta[offset + i * index_step]) * x[n - i]
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        for k in range(min(n_orig - n - 1, (nwin - offset) // index_step)):
            y[t] += (win[offset + k * index_step] + eta * delta[offset + k * index_step]) * x[n + k + 1]
    return y


@pytest.mark.parametrize("sr_orig,sr_new,n", [(17000.0, 48000.0, 90), (96000.0, 48000.0, 300), (44100.0, 48000.0, 101),
                                               (52345.678, 48000.0, 150), (48000.0, 16000.0, 90)])
def test_vectorised_equals_the_per_sample_loop_bit_for_bit(sr_orig, sr_new, n):
    x = np.random.default_rng(int(sr_orig)).standard_normal(n)
    y = rs.resample(x, sr_orig, sr_new)
    assert y.shape == (int(n * sr_new / sr_orig),)
    assert np.array_equal(y, loop_resample(x, sr_orig, sr_new))


def test_filter_design_values():
    win, num_table = rs.kaiser_best()
    assert num_table == 512 and win.shape == (64 * 512 + 1,)
    assert win[0] == rs.KAISER_BEST["rolloff"]                      # sinc(0) * kaiser centre (= 1)
    assert abs(win[-1]) < 1e-7 and np.all(np.abs(win[512::512]) < 0.06)  # near the zero crossings of the rolled-off sinc
    # unit DC gain when upsampling: the taps seen by one output sample sum to ~1
    assert abs(win[::512].sum() * 2 - win[0] - 1.0) < 1e-4


@pytest.mark.parametrize("sr_orig,sr_new", [(17000.0, 48000.0), (31234.5, 48000.0), (96000.0, 48000.0), (130000.0, 48000.0)])
def test_band_limited_signals_come_through(sr_orig, sr_new):
    n = 6000
    t_in = np.arange(n) / sr_orig
    f = [200.0, 1000.0, 0.35 * min(sr_orig, sr_new)]
    x = sum(np.sin(2 * np.pi * fk * t_in + k) for k, fk in enumerate(f))
    y = rs.resample(x, sr_orig, sr_new)
    t_out = np.arange(y.shape[0]) / sr_new
    want = sum(np.sin(2 * np.pi * fk * t_out + k) for k, fk in enumerate(f))
    edge = int(80 * max(sr_new / sr_orig, 1.0) * max(sr_orig / sr_new, 1.0)) + 10  # the filter's 64 zero crossings
    # (non-integer decimation: the table step int(scale * 512) is truncated, a gain error of ~1.6e-4 per component)
    assert np.max(np.abs(y[edge:-edge] - want[edge:-edge])) < (1e-3 if sr_new < sr_orig else 2e-4)


def test_content_above_the_new_nyquist_is_removed():
    sr_orig, sr_new, n = 120000.0, 48000.0, 8000
    t_in = np.arange(n) / sr_orig
    x = np.sin(2 * np.pi * 30000.0 * t_in)       # above 24 kHz
    y = rs.resample(x, sr_orig, sr_new)
    assert np.max(np.abs(y[300:-300])) < 1e-3    # > 60 dB down


def test_axis_dtype_and_errors():
    x = np.random.default_rng(3).standard_normal((3, 200)).astype(np.float32)
    y = rs.resample(x, 20000.0, 48000.0)
    assert y.dtype == np.float32 and y.shape == (3, 480)
    y0 = rs.resample(x.T.copy(), 20000.0, 48000.0, axis=0)
    assert np.array_equal(y0, y.T)
    for r in range(3):
        assert np.array_equal(y[r], rs.resample(x[r], 20000.0, 48000.0))
    assert rs.resample(np.arange(10), 1.0, 2.0).dtype == np.float64
    with pytest.raises(ValueError):
        rs.resample(np.zeros(1), 48000.0, 1000.0)
    with pytest.raises(ValueError):
        rs.resample(np.zeros(10), 0.0, 1000.0)
