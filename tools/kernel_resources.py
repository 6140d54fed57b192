#!/usr/bin/env python3
"""Compile the HIP engine with -Rpass-analysis=kernel-resource-usage and print one line per kernel."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "pffdtd_amd" / "csrc"
pat = sys.argv[1] if len(sys.argv) > 1 else ""
out = ""
for unit in ("pf_engine.hip", "pf_engine_f64.hip"):  # (the fp32 and the fp64 instantiation of the engine)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-ffp-contract=off",
           "-Wno-unused-value", "-I", str(ROOT / "include"), "-I", str(CSRC), str(CSRC / unit), "-o",
           "/tmp/pf_res.o", "-Rpass-analysis=kernel-resource-usage"]
    out += subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:Function Name: (\S+)|\s*(\w[\w \[\]/]*): (\d+))", line)
    if not m:
        continue
    if m.group(1):
        cur = m.group(1)
        rows[cur] = {}
    elif cur:
        rows[cur][m.group(2).strip()] = int(m.group(3))
if not rows:
    print(out[-3000:])
    sys.exit("compile failed or no kernels found")
dem = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>7} {'occ':>4} {'LDS':>6}  kernel")
for (k, v), d in zip(rows.items(), dem):
    d = re.sub(r"\(.*", "", d).replace("void pf::", "")
    if pat and not re.search(pat, d):
        continue
    print(f"{v.get('VGPRs', 0):5d} {v.get('AGPRs', 0):5d} {v.get('TotalSGPRs', 0):5d} {v.get('ScratchSize [bytes/lane]', 0):7d} "
          f"{v.get('Occupancy [waves/SIMD]', 0):4d} {v.get('LDS Size [bytes/block]', 0):6d}  {d}")
