#!/usr/bin/env python3
"""Generate the golden receiver outputs under tests/golden/ by running the COMPILED REFERENCE
(oracle/_ref/fdtd_main_cpu_{single,double}.x, built by oracle/Makefile from /root/reference/c_cuda) on the
synthetic scenes of tests/cases.py.  Runs only where /root/reference exists (the build container).

Each fixture = {u_out: what the reference wrote to sim_outs.h5, digest: sha256 of the input arrays} -- data
only; the scenes are regenerated from tests/cases.py by the tests, the digest catches generator drift.
"""
import hashlib
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]
import cases  # noqa: E402
import oracle  # noqa: E402
from pffdtd_amd import synth  # noqa: E402

GOLDEN_CASES = ["cart_lossy", "cart_mb11", "cart_outside", "cart_outside_oddz", "cart_wall2", "fcc1_lossy",
                "fcc1_outside", "fcc2_lossy", "fcc2_outside", "fcc2_mb11"]


def digest(sim):
    h = hashlib.sha256()
    for f in sorted(sim):
        for k in sorted(sim[f]):
            a = np.ascontiguousarray(sim[f][k])
            h.update(f"{f}/{k}/{a.dtype}/{a.shape}".encode())
            h.update(a.tobytes())
    return h.hexdigest()


def main():
    for name in GOLDEN_CASES:
        sim = cases.make_sim(name)
        for prec in ("double", "single"):
            with tempfile.TemporaryDirectory() as d:
                synth.write_folder(sim, d)
                u_out, _ = oracle.run_reference(d, prec, threads=4)
            out = HERE / f"{name}_{prec}.npz"
            np.savez_compressed(out, u_out=u_out, digest=np.array(digest(sim)))
            print(f"{out.name}: {u_out.shape} peak {np.abs(u_out).max():.6e}")


if __name__ == "__main__":
    main()
