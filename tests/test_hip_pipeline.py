"""GPU: the whole chain absorption data + scene export -> RIRs (examples/ctk_rir.py) on a small CTK configuration."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_ctk_export_to_rir(tmp_path):
    spec = importlib.util.spec_from_file_location("ctk_rir", ROOT / "examples" / "ctk_rir.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    po = mod.main(["--out", str(tmp_path), "--fmax", "300", "--ppw", "7", "--duration", "0.12", "--air_abs", "stokes"])
    r = np.atleast_2d(po.r_out_f)
    assert r.shape[0] == 6 and abs(po.Fs_f - 48000.0) < 0.01 and np.isfinite(r).all()
    e = np.cumsum(r[:, ::-1] ** 2, axis=1)[:, ::-1]  # Schroeder integral: sound decays after the direct part
    n = r.shape[1]
    assert np.all(e[:, n // 2] < 0.5 * e[:, n // 10]) and np.abs(r).max() > 0
    assert (tmp_path / "sim" / "sim_outs_processed.h5").exists() and (tmp_path / "sim" / "sim_outs.h5").exists()
