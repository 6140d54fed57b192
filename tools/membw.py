#!/usr/bin/env python3
"""HBM streaming ceilings on this GPU with torch elementwise kernels (calibration for the roofline discussion)."""
import torch
n = 1024 ** 3
a = torch.rand(n, device="cuda"); b = torch.rand(n, device="cuda"); c = torch.empty(n, device="cuda")
def bench(fn, nbytes, name, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / it
    print(f"{name:28s} {ms:8.3f} ms  {nbytes/ms/1e6:8.1f} GB/s", flush=True)
bench(lambda: c.copy_(a), 8 * n, "copy 1R+1W (fp32)")
bench(lambda: torch.add(a, b, out=c), 12 * n, "add  2R+1W (fp32)")
bench(lambda: torch.add(a, c, out=c), 12 * n, "add  2R+1W in-place (u0 rw)")
bench(lambda: c.fill_(1.0), 4 * n, "fill 1W")
bench(lambda: a.sum(), 4 * n, "sum  1R")
bench(lambda: torch.dot(a, b), 8 * n, "dot  2R")
