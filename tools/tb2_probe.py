#!/usr/bin/env python3
"""Research probe: two fused time steps per pass (temporal blocking) of the pure 7-point air update vs two passes of
the production single-step kernel, on a free-field grid.  Validates bit-equality on the box [m, N-m)^3 and times it."""
import functools
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pffdtd_amd import engine, sim_data, synth  # noqa: E402

print = functools.partial(print, flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = 8
from pffdtd_amd import build  # noqa: E402
L = build.load_probe()
sim = synth.shoebox(n, n, n, Nt=8, box=False, lossy=False)
sd = sim_data.SimData.from_sim(sim, "single", build_mask=False)
sd.scale_input()
sd.in_sigs[:] = 0  # no source: pure propagation of the random field
P = engine.grid_pitch(n, 4)
g = [torch.zeros((n, n * P), dtype=torch.float32, device="cuda") for _ in range(4)]
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
A0 = (torch.rand(g[0].shape, generator=gen, device="cuda") * 2 - 1) * 1e-3
B0 = (torch.rand(g[0].shape, generator=gen, device="cuda") * 2 - 1) * 1e-3
g[0].copy_(A0); g[1].copy_(B0)
eng = engine.HipEngine(sd, ext_u0=g[0].data_ptr(), ext_u1=g[1].data_ptr(), timing=True)
eng.run(0, 2)
eng.sync()
t0 = time.perf_counter(); eng.run(2, 6); eng.sync(); t_single = (time.perf_counter() - t0) / 6
tm = eng.timing()
print(f"single-step engine: {t_single*1e3:.3f} ms/step (air kernel {tm['air_ms_total']/tm['air_launches']:.3f} ms)")
# reference for the fused pass: two engine steps from (A0, B0): after 2 steps u1 = u^{n+2} (in g[1]'s storage), u0 = u^{n+1}
g[0].copy_(A0); g[1].copy_(B0)
eng2 = engine.HipEngine(sd, ext_u0=g[0].data_ptr(), ext_u1=g[1].data_ptr())
eng2.run(0, 2); eng2.sync()
ref_n1 = g[0].clone().view(n, n, P)   # u^{n+1}: written into u0's array at step 0, then became u1, ... after 2 rotations
ref_n2 = g[1].clone().view(n, n, P)
# engine state after 2 steps: u1 = u^{n+2}; which storage? step0 writes n+1 into g[0]; step1 writes n+2 into g[1]
g[0].copy_(A0); g[1].copy_(B0); g[2].zero_(); g[3].zero_()
for tye in [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "204,304,404,308,302,408".split(","))]:
    for chunk in [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "16,32,64".split(","))]:
        ms = L.pf_tb2_probe(g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), n, n, n, float(sd.a1), float(sd.a2), m, tye, chunk, 5)
        if ms < 0:
            print("probe failed:", L.pf_probe_last_error().decode()); continue
        C = g[2].view(n, n, P)[m:n - m, m:n - m, m:n - m]
        D = g[3].view(n, n, P)[m:n - m, m:n - m, m:n - m]
        okC = bool(torch.equal(C, ref_n1[m:n - m, m:n - m, m:n - m]))
        okD = bool(torch.equal(D, ref_n2[m:n - m, m:n - m, m:n - m]))
        cells = (n - 2 * m) ** 3
        print(f"tb2 tye={tye} chunk={chunk}: {ms:.3f} ms per 2 steps = {ms/2:.3f} ms/step-equivalent on {cells/n**3*100:.0f}% of the grid; "
              f"bit-equal C {okC} D {okD}; vs 2 x air kernel {2*tm['air_ms_total']/tm['air_launches']:.3f} ms")
