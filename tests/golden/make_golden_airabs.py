#!/usr/bin/env python3
"""Golden vectors of the reference's air-absorption package (python/air_abs) for tests/test_air_abs.py: the ISO 9613
coefficient table and the three filters on a seeded two-channel decaying-noise response.  Build container only;
numba is replaced by an identity decorator (the reference's jit functions are plain numpy underneath)."""
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
np.float = float
nb = types.ModuleType("numba"); nb.jit = lambda *a, **k: (lambda f: f); nb.prange = range; sys.modules["numba"] = nb
sys.path.insert(0, "/root/reference/python")
from air_abs.get_air_absorption import get_air_absorption  # noqa: E402
from air_abs.visco_filter import apply_visco_filter  # noqa: E402
from air_abs.modal_filter import apply_modal_filter  # noqa: E402
from air_abs.ola_filter import apply_ola_filter  # noqa: E402

out = {}
f = np.logspace(0, np.log10(80e3), 40)
for i, (Tc, rh) in enumerate(((20, 50), (10, 15), (35.5, 80))):
    rd = get_air_absorption(f, Tc, rh)
    out[f"coef{i}_TcRh"] = np.array([Tc, rh], dtype=np.float64)
    for k, v in rd.items():
        out[f"coef{i}_{k}"] = np.asarray(v, dtype=np.float64)
out["coef_f"] = f
rng = np.random.default_rng(11)
Fs = 48000.0
x = rng.standard_normal((2, 2500)) * np.exp(-np.arange(2500) / 600.0)
out["x"], out["Fs"] = x, np.float64(Fs)
out["visco"] = apply_visco_filter(x, Fs, 20, 50)
out["visco_1ch_NdB80"] = apply_visco_filter(x[0], Fs, 10, 30, NdB=80)
out["modal"] = apply_modal_filter(x, Fs, 20, 50)
out["modal_pad"] = apply_modal_filter(x[1], Fs, 25, 40, pad_t=0.004)
out["ola"] = apply_ola_filter(x, Fs, 20, 50)
out["ola_nw256"] = apply_ola_filter(x[0], 25486.6, 15, 60, Nw=256)
np.savez_compressed(HERE / "airabs_reference.npz", **out)
print("wrote airabs_reference.npz", {k: v.shape for k, v in out.items() if k in ("visco", "modal", "ola", "modal_pad", "ola_nw256")})
