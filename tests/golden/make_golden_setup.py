#!/usr/bin/env python3
"""Golden vectors of the reference's setup-side math (SimConsts, SimComms) for tests/test_setup_io.py.
The reference modules are imported from /root/reference behind a test-only empty h5py module (this container only)."""
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
np.float = float
sys.modules.setdefault("h5py", types.ModuleType("h5py"))
sys.path.insert(0, "/root/reference/python")
from fdtd.sim_consts import SimConsts  # noqa: E402
from fdtd.sim_comms import SimComms  # noqa: E402

out = {}
for tag, fcc in (("cart", False), ("fcc", True)):
    sc = SimConsts(Tc=20, rh=50, fmax=700.0, PPW=8.0, fcc=fcc)
    for k in ("h", "c", "Ts", "SR", "l", "l2"):
        out[f"{tag}_{k}"] = np.float64(getattr(sc, k))
    N = (20, 18, 16)
    xv, yv, zv = (np.arange(n) * sc.h + o for n, o in zip(N, (-0.3, 0.1, 0.05)))
    cm = SimComms.__new__(SimComms)
    cm.h, cm.Ts, cm.l2, cm.fcc_flag, cm.fcc = sc.h, sc.Ts, sc.l2, int(fcc), fcc
    cm.xv, cm.yv, cm.zv, cm._diff = xv, yv, zv, False
    S = np.array([xv[7] + 0.37 * sc.h, yv[9] - 0.21 * sc.h, zv[6] + 0.5 * sc.h])
    R = np.array([[xv[11] - 0.1 * sc.h, yv[5] + 0.45 * sc.h, zv[9] + 0.3 * sc.h], [xv[5] + 0.6 * sc.h, yv[12] - 0.4 * sc.h, zv[4] + 0.25 * sc.h]])
    cm.prepare_source_pts(S)
    cm.prepare_receiver_pts(R)
    cm.prepare_source_signals(60 * sc.Ts, sig_type="dhann30")
    out[f"{tag}_S"], out[f"{tag}_R"], out[f"{tag}_xv0"] = S, R, np.array([xv[0], yv[0], zv[0]])
    out[f"{tag}_in_alpha"], out[f"{tag}_in_ixyz"] = cm.in_alpha, cm.in_ixyz
    out[f"{tag}_out_alpha"], out[f"{tag}_out_ixyz"] = cm.out_alpha, cm.out_ixyz
    out[f"{tag}_in_sigs"] = cm.in_sigs.copy()
    cm.diff_source()
    out[f"{tag}_in_sigs_diff"] = cm.in_sigs.copy()
np.savez_compressed(HERE / "setup_reference.npz", **out)
print("wrote", HERE / "setup_reference.npz", len(out), "arrays")

# ---- post-processing (python/fdtd/process_outputs.py) on a synthetic receiver set ----
for m in ("resampy", "matplotlib", "matplotlib.pyplot"):
    sys.modules.setdefault(m, types.ModuleType(m))
sys.modules["resampy"].resample = lambda *a, **k: None
air = types.ModuleType("air_abs"); sys.modules.setdefault("air_abs", air)
for sub, fn in (("visco_filter", "apply_visco_filter"), ("modal_filter", "apply_modal_filter"), ("ola_filter", "apply_ola_filter")):
    mod = types.ModuleType(f"air_abs.{sub}"); setattr(mod, fn, None); sys.modules.setdefault(f"air_abs.{sub}", mod)


class _FakeH5:  # initial_process() stores r_out into sim_outs.h5 midway (process_outputs.py:97-104): swallow that
    def __init__(self, *a, **k): pass
    def __delitem__(self, k): raise KeyError(k)
    def create_dataset(self, *a, **k): pass
    def close(self): pass


sys.modules["h5py"].File = _FakeH5
from fdtd.process_outputs import ProcessOutputs  # noqa: E402

rng = np.random.default_rng(5)
Ts = 1 / 25000.0
u_out = rng.standard_normal((16, 400)) * np.exp(-np.arange(400) / 80.0)
alpha = rng.random((2, 8)); alpha /= alpha.sum(axis=1, keepdims=True)
post = {"post_u_out": u_out, "post_alpha": alpha, "post_Ts": np.float64(Ts)}
for diff in (True, False):
    po = ProcessOutputs.__new__(ProcessOutputs)
    po.u_out, po.out_alpha, po.data_dir, po.diff, po.Ts = u_out, alpha, Path("/nonexistent"), diff, Ts
    po.Ts_f, po.Fs_f, po.Fs, po.Nt, po.Nt_f = Ts, 1 / Ts, 1 / Ts, 400, 400
    po.initial_process(fcut=10.0, N_order=4)
    r_out = po.r_out
    post[f"post_r_out_f_diff{int(diff)}"] = po.r_out_f.copy()
    po.apply_lowpass(fcut=4000.0, N_order=8, symmetric=True)
    post[f"post_lowpass_diff{int(diff)}"] = po.r_out_f.copy()
post["post_r_out"] = r_out
np.savez_compressed(HERE / "post_reference.npz", **post)
print("wrote post_reference.npz")
