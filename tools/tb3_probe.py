#!/usr/bin/env python3
"""Research probe (round 5): THREE fused time steps per pass of the pure 7-point air update (the product kernel k_tb3, pffdtd_amd/csrc/pf_tb3.h) against
three passes of the production single-step kernel and against the production two-steps-per-pass kernel, on a free-field grid.
Validates bit-equality on the box [m, N-m)^3 and times it.   usage: tb3_probe.py [n] [variants] [chunks]"""
import functools
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pffdtd_amd import build, engine, sim_data, synth  # noqa: E402

print = functools.partial(print, flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = 8
L = build.load_probe()
import ctypes
vp, i32, i64, d = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
L.pf_tb3_probe.restype = d
L.pf_tb3_probe.argtypes = [vp, vp, vp, vp, i64, i64, i64, d, d, i32, i32, i32, i32]
sim = synth.shoebox(n, n, n, Nt=8, box=False, lossy=False)
sd = sim_data.SimData.from_sim(sim, "single", build_mask=False)
sd.scale_input()
sd.in_sigs[:] = 0  # no source: pure propagation of the random field
P = engine.grid_pitch(n, 4)
g = [torch.zeros((n, n * P), dtype=torch.float32, device="cuda") for _ in range(5)]
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
A0 = (torch.rand(g[0].shape, generator=gen, device="cuda") * 2 - 1) * 1e-3
B0 = (torch.rand(g[0].shape, generator=gen, device="cuda") * 2 - 1) * 1e-3
# reference: three single steps of the engine from (A0, B0): step 0 writes u^{n+1} into g[0], step 1 u^{n+2} into g[1], step 2 u^{n+3} into g[0]
g[0].copy_(A0); g[1].copy_(B0)
eng = engine.HipEngine(sd, ext_u0=g[0].data_ptr(), ext_u1=g[1].data_ptr(), air_variant=25)
eng.run(0, 2); eng.sync()
ref_n2 = g[1].clone().view(n, n, P)
eng.run(2, 1); eng.sync()
ref_n3 = g[0].clone().view(n, n, P)
eng.close()
g[0].copy_(A0); g[1].copy_(B0)
ms2 = L.pf_tb2_probe(g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), n, n, n, float(sd.a1), float(sd.a2), m, 10304, 16, 5)
print(f"k_tb2_reg<3,4> banded chunk 16: {ms2:.3f} ms per 2 steps = {ms2/2:.3f} ms/step")
cells = (n - 2 * m) ** 3
for var in [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "308,10308,208,304,404,212,306".split(","))]:
    for chunk in [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "16,32,64".split(","))]:
        g[3].zero_(); g[4].zero_()
        ms = L.pf_tb3_probe(g[0].data_ptr(), g[1].data_ptr(), g[3].data_ptr(), g[4].data_ptr(), n, n, n, float(sd.a1), float(sd.a2), m, var, chunk, 5)
        if ms < 0:
            print("probe failed:", L.pf_probe_last_error().decode()); continue
        Dv = g[3].view(n, n, P)[m:n - m, m:n - m, m:n - m]
        Ev = g[4].view(n, n, P)[m:n - m, m:n - m, m:n - m]
        okD = bool(torch.equal(Dv, ref_n2[m:n - m, m:n - m, m:n - m]))
        okE = bool(torch.equal(Ev, ref_n3[m:n - m, m:n - m, m:n - m]))
        print(f"tb3 variant={var} chunk={chunk}: {ms:.3f} ms per 3 steps = {ms/3:.3f} ms/step-equivalent on {cells/n**3*100:.0f}% of the grid "
              f"({cells*16/ms/1e6:.0f} GB/s of compulsory traffic); bit-equal u^(n+2) {okD} u^(n+3) {okE}")
