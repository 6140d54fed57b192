"""Air-absorption post-filters against values captured from the reference's python/air_abs package
(tests/golden/airabs_reference.npz, made by make_golden_airabs.py)."""
from pathlib import Path

import numpy as np
import pytest

from pffdtd_amd import air_abs

G = np.load(Path(__file__).resolve().parent / "golden" / "airabs_reference.npz")
TOL = dict(rtol=1e-13, atol=1e-15)  # same numpy expressions; scipy.fft may differ in the last bits across builds


@pytest.mark.parametrize("i", [0, 1, 2])
def test_iso9613_coefficients(i):
    Tc, rh = G[f"coef{i}_TcRh"]
    rd = air_abs.get_air_absorption(G["coef_f"], Tc, rh)
    keys = [k[len(f"coef{i}_"):] for k in G.files if k.startswith(f"coef{i}_") and not k.endswith("TcRh")]
    assert len(keys) == 17
    for k in keys:
        assert np.array_equal(np.asarray(rd[k], dtype=np.float64), G[f"coef{i}_{k}"]), k


def test_range_checks():
    with pytest.raises(ValueError):
        air_abs.get_air_absorption(1000.0, 60, 50)
    with pytest.raises(ValueError):
        air_abs.get_air_absorption(1000.0, 20, 5)


def test_stokes_filter():
    x, Fs = G["x"], float(G["Fs"])
    y = air_abs.apply_visco_filter(x, Fs, 20, 50)
    assert y.shape == G["visco"].shape and np.array_equal(y, G["visco"])
    y1 = air_abs.apply_visco_filter(x[0], Fs, 10, 30, NdB=80)
    assert y1.ndim == 1 and np.array_equal(y1, G["visco_1ch_NdB80"])


def test_modal_filter():
    x, Fs = G["x"], float(G["Fs"])
    y = air_abs.apply_modal_filter(x, Fs, 20, 50)
    assert y.shape == G["modal"].shape
    np.testing.assert_allclose(y, G["modal"], **TOL)
    np.testing.assert_allclose(air_abs.apply_modal_filter(x[1], Fs, 25, 40, pad_t=0.004), G["modal_pad"], **TOL)


def test_ola_filter():
    x, Fs = G["x"], float(G["Fs"])
    y = air_abs.apply_ola_filter(x, Fs, 20, 50)
    assert y.shape == G["ola"].shape
    np.testing.assert_allclose(y, G["ola"], **TOL)
    np.testing.assert_allclose(air_abs.apply_ola_filter(x[0], 25486.6, 15, 60, Nw=256), G["ola_nw256"], **TOL)


def test_filters_attenuate_high_frequencies_more():
    """Physics sanity: late, high-frequency content loses the most energy."""
    Fs = 48000.0
    n = np.arange(4096)
    lo, hi = np.sin(2 * np.pi * 500 * n / Fs), np.sin(2 * np.pi * 15000 * n / Fs)
    for fn in (air_abs.apply_modal_filter, air_abs.apply_ola_filter):
        a, b = fn(lo, Fs, 20, 50)[3000:4000], fn(hi, Fs, 20, 50)[3000:4000]
        assert np.sqrt(np.mean(b ** 2)) < 0.9 * np.sqrt(np.mean(a ** 2))
