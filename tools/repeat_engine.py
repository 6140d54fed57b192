#!/usr/bin/env python3
"""(Superseded by tools/placement_probe.py, which found the cause: grid placement.)
Does the pair kernel's speed depend on which engine of a process it runs in?  Builds the bench scene once, then creates,
runs and destroys the engine several times in one process and prints the blocked kernel's time per launch for each."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from pffdtd_amd import dist as pdist  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
K, W = 40, 8
sd = bench.build_scene(n, K + W, "single", False, True, 11)
gen = torch.Generator(device="cuda")
for r in range(reps):
    gen.manual_seed(1234)
    runner, loc, info = pdist.make_hip_runner(sd, 0, 1, 0, None, timing=True)
    for g in runner.st.grids:
        g.copy_((torch.rand(g.shape, generator=gen, device=g.device, dtype=torch.float32) * 2.0 - 1.0) * 1e-3)
    torch.cuda.synchronize()
    eng = runner.st.eng
    eng.run(0, W); eng.sync(); eng.timing(reset=True)
    t0 = time.perf_counter(); eng.run(W, K); eng.sync(); el = time.perf_counter() - t0
    tm = eng.timing()
    print(f"engine {r}: {el / K * 1e3:.4f} ms/step, pair kernel {tm['tb2_ms_total'] / max(tm['tb2_launches'], 1):.4f} ms/launch, "
          f"grids at {[hex(g.data_ptr()) for g in runner.st.grids]}", flush=True)
    runner.st.close()
    runner.st.grids.clear()
    del runner, eng
    torch.cuda.empty_cache()
