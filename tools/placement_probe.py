#!/usr/bin/env python3
"""Is the pair kernel's speed a property of WHERE its four grids were allocated?  Three engines of the bench scene are kept
alive side by side (debug 0x8000: no creation-time measurement) and timed in turn, twice round: an allocation effect shows
as engine-specific times that repeat, a power / thermal effect as times that follow the order of the runs."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from pffdtd_amd import dist as pdist  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
neng = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
fcc = len(sys.argv) > 4 and sys.argv[4] == "fcc"
K, W = 20, 4
sd = bench.build_scene(n, (K + W) * (rounds + 1), "single", fcc, True, 11)
gen = torch.Generator(device="cuda")
engs = []
for r in range(neng):
    gen.manual_seed(1234)
    runner, loc, info = pdist.make_hip_runner(sd, 0, 1, 0, None, timing=True)
    for g in runner.st.grids:
        g.copy_((torch.rand(g.shape, generator=gen, device=g.device, dtype=torch.float32) * 2.0 - 1.0) * 1e-3)
    torch.cuda.synchronize()
    engs.append(runner)
    print(f"engine {r}: placement {runner.st.eng.timing()['place_ms']}, grids at {[hex(g.data_ptr()) for g in runner.st.grids]}", flush=True)
for rd in range(rounds):
    for r, runner in enumerate(engs):
        eng = runner.st.eng
        n0 = rd * (K + W)
        eng.run(n0, W); eng.sync(); eng.timing(reset=True)
        t0 = time.perf_counter(); eng.run(n0 + W, K); eng.sync(); el = time.perf_counter() - t0
        tm = eng.timing()
        print(f"round {rd} engine {r}: {el / K * 1e3:.4f} ms/step, pair kernel {tm['tb2_ms_total'] / max(tm['tb2_launches'], 1):.4f} ms/launch", flush=True)
