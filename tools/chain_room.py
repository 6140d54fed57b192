#!/usr/bin/env python3
"""One of the reference's rooms through the multi-device chain object (pf_multi_*), cut along x (the reference's arrangement) and
along file z (slab engines store the axes exchanged), on however many devices are visible -- on a 1-GPU box as virtual slabs
(both slabs on device 0: a comparison of the two cuts, not a scaling figure).

    python tools/chain_room.py mv_fcc_gpu --slabs 2 --steps 100
"""
import argparse
import json
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from pffdtd_amd import engine, scenes, sim_data  # noqa: E402
from pffdtd_amd.dist import _DevMem  # noqa: E402
from pffdtd_amd.sim_setup import sim_setup  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("name", choices=sorted(scenes.CONFIGS))
ap.add_argument("--slabs", type=int, default=2)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--transport", default="auto", choices=["auto", "peer", "rccl"])
a = ap.parse_args()

tmp = tempfile.TemporaryDirectory()
base = Path(tmp.name)
mats = scenes.write_materials(base / "materials")
folder = base / a.name
sim_setup(**scenes.setup_kwargs(a.name, folder, mats, save_folder_gpu=folder, compress=0))
ndev = engine.device_count()
devices = list(range(a.slabs)) if ndev >= a.slabs else [i % ndev for i in range(a.slabs)]
tr = {"auto": engine.PF_TRANSPORT_AUTO, "peer": engine.PF_TRANSPORT_PEER, "rccl": engine.PF_TRANSPORT_RCCL}[a.transport]
for label, flags in (("cut along x (reference)", engine.PF_MULTI_CUT_X), ("cut along file z, axes exchanged", engine.PF_MULTI_CUT_Z)):
    sd = sim_data.SimData.from_folder(folder, "single", build_mask=False)
    sd.scale_input()
    m = engine.HipMulti(sd, devices, multi_flags=flags, transport=tr, verify_exchange=4)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    for g in range(a.slabs):
        sl = m.slab(g)
        (nx, ny, _), pitch, _ = sl["engine"].layout()
        with torch.cuda.device(sl["device"]):
            for p in sl["engine"].state_grids():
                t = torch.as_tensor(_DevMem(p, (nx, ny * pitch), "<f4"), device=f"cuda:{sl['device']}")
                t.copy_((torch.rand(t.shape, generator=gen, device=t.device) * 2 - 1) * 1e-3)
            torch.cuda.synchronize()
    m.run(0, a.warmup)
    t0 = time.perf_counter()
    m.run(a.warmup, a.steps)
    el = time.perf_counter() - t0
    info = m.info()
    print(json.dumps({"config": a.name, "cut": label, "slabs": a.slabs, "devices": devices, "virtual": ndev < a.slabs,
                      "grid": [sd.Nx, sd.Ny, sd.Nz], "ms_per_step": round(el / a.steps * 1e3, 4),
                      "gvox_per_s": round(sd.Npts * a.steps / el / 1e9, 2), "transport": info["transport_name"],
                      "exchange_verified": info["exchange_verified"], "cut_along_z": info["cut_along_z"],
                      "ranges": [[m.slab(g)["x0"], m.slab(g)["x1"]] for g in range(a.slabs)]}), flush=True)
    m.close()
