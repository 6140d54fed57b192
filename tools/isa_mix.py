#!/usr/bin/env python3
"""Instruction mix of the kernels in an AMDGPU assembly listing (hipcc -S --cuda-device-only): per kernel whose name contains
the given substring, the count of vector / scalar / memory / LDS instructions by opcode.  usage: isa_mix.py file.s substring"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r"\n(?=_Z[^\n:]*:[^\n]*\n)", txt)
for f in parts:
    name = f.split(":")[0]
    if not name.startswith("_Z") or want not in name:
        continue
    c = collections.Counter()
    for line in f.split("\n"):
        line = line.strip()
        m = re.match(r"((?:v|s|ds|global|buffer|scratch|flat)_[a-z0-9_]+)", line)
        if m:
            op = m.group(1)
            if "dpp" in line and op.startswith("v_"):
                op += "(dpp)"
            c[op] += 1
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    salu = sum(v for k, v in c.items() if k.startswith("s_"))
    print(f"{name[:90]}\n   VALU {valu}  SALU {salu}  of {sum(c.values())}")
    print("   " + ", ".join(f"{k}:{v}" for k, v in c.most_common(28)))
