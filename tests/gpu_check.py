#!/usr/bin/env python3
"""Verbose GPU parity diagnostics (prints per-case max differences instead of stopping at the first).
Part of the test infrastructure (uses the CPU oracle); run as `python tests/gpu_check.py [variants]`."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "oracle"), str(ROOT / "tests")]  # test infrastructure: may use the oracle
import cases  # noqa: E402
import oracle  # noqa: E402
from pffdtd_amd import engine  # noqa: E402

import time, functools
print = functools.partial(print, flush=True)
T0 = time.time()
oracle.lib().oracle_set_threads(8)
print("devices:", engine.device_count(), engine.lib().pf_version())
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 3, 4, 7, 25, 41, 256, 259]
bad = 0
for name in cases.CASES:
    for prec in ("double", "single"):
        sd = cases.make_sd(name, prec)
        print(f"[{time.time()-T0:.1f}s] {name} {prec}: oracle...")
        e = oracle.Engine(sd)
        for n in range(sd.Nt):
            e.step(n)
        ref_u1 = e.grid(1).copy()
        ref_out = sd.u_out.copy()
        e.close()
        print(f"[{time.time()-T0:.1f}s] oracle done")
        for v in variants:
            sd.u_out[:] = 0
            try:
                eng = engine.HipEngine(sd, air_variant=v, air_chunk=7)
                eng.run(0, sd.Nt)
                u1 = eng.get_grid(1)
                eng.close()
            except Exception as ex:  # noqa: BLE001
                print(f"{name:12s} {prec:6s} v{v}: EXC {ex}")
                bad += 1
                continue
            d_out = np.abs(sd.u_out - ref_out).max()
            I = (slice(1, -1),) * 3
            d_u1 = np.abs(u1[I] - ref_u1[I]).max()
            nbad = int((u1[I] != ref_u1[I]).sum())
            ok = d_out == 0 and d_u1 == 0
            bad += not ok
            msg = "" if ok else f" first bad idx {np.argwhere(u1[I] != ref_u1[I])[:4].tolist()}"
            print(f"{name:12s} {prec:6s} v{v}: out {d_out:.3e} u1 {d_u1:.3e} nbad {nbad} peak {np.abs(ref_out).max():.3e} "
                  f"{'OK' if ok else 'MISMATCH'}{msg}")
print("TOTAL MISMATCH:", bad)
