#!/usr/bin/env python3
"""A shoebox stepped as ONE domain and as a chain of virtual slabs on this device (pairs, wall regions in the slabs, every exchange
checked): receivers must agree bit for bit.  A stress check of the chain's stream choreography at sizes the oracle is too slow for.
usage: tools/chain_vs_single.py [Nx] [slabs] [steps] [Ny] [Nz]   (Nz = 248 k + 32 leaves column strips the pencils fit)"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pffdtd_amd import engine, sim_data, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 288
G = int(sys.argv[2]) if len(sys.argv) > 2 else 3
K = int(sys.argv[3]) if len(sys.argv) > 3 else 700
ny = int(sys.argv[4]) if len(sys.argv) > 4 else 200
nz = int(sys.argv[5]) if len(sys.argv) > 5 else 280
rcv = [[n // 3, ny // 2, nz // 2 + 3], [5, 6, 7], [n - 9, ny - 10, nz // 3], [n // 2, 4, nz // 2], [n // 2 + 1, ny // 2, nz - 7], [n // G + 1, 9, 11], [n // G - 2, ny - 12, 13]]
sim = synth.shoebox(n, ny, nz, Nt=K, Nm=2, Mb=[11, 3], src=[n // 2 + 7, 41, 47], rcv=rcv)  # (the source near two walls: reflections from step ~60)
outs = {}
for label in ("single", "chain", "chain, single-step shell"):
    sd = sim_data.SimData.from_sim(sim, "single", build_mask=False)
    sd.scale_input()
    if label == "single":
        e = engine.HipEngine(sd, timing=True, air_variant=40)
        e.run(0, K)
        info = e.timing()["wall_blocks"]
        e.close()
    else:
        m = engine.HipMulti(sd, [0] * G, multi_flags=engine.PF_MULTI_FORCE_PAIRS, air_variant=40, verify_exchange=K, debug=0x10000000 if "single-step" in label else 0)
        m.run(0, K)
        info = (m.info()["exchange_verified"], [sum(m.slab(g)["engine"].timing()["wall_blocks"]) for g in range(G)], [m.slab(g)["paired"] for g in range(G)])
        m.close()
    outs[label] = sd.u_out.copy()
    print(label, info, "max |u_out| %.3e" % np.abs(sd.u_out).max(), flush=True)
ok = all(np.array_equal(outs["single"], v) for v in outs.values())
print("receivers identical" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
