#!/usr/bin/env python3
"""HBM rate of linear streams by read:write mix (torch elementwise kernels, 1024^3 fp32 operands): the pair kernel moves
2 reads + 2 writes per cell, the single-step kernels 2 reads + 1 write."""
import torch
n = 1024 ** 3
a, b, c, d = (torch.rand(n, device="cuda") for _ in range(4))
def t(fn, nbytes, label, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{label:28s} {ms:7.3f} ms  {nbytes/ms/1e9:6.2f} TB/s", flush=True)
t(lambda: torch.add(a, b, out=c), 12 * n, "add   2R+1W (33% writes)")
t(lambda: c.copy_(a), 8 * n, "copy  1R+1W (50% writes)")
t(lambda: torch.mul(a, 2.0, out=c), 8 * n, "scale 1R+1W (50% writes)")
t(lambda: c.fill_(1.0), 4 * n, "fill  0R+1W")
t(lambda: a.sum(), 4 * n, "sum   1R+0W")
t(lambda: torch.addcmul(a, b, c, out=d), 16 * n, "addcmul 3R+1W (25% writes)")
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    with torch.cuda.stream(s0): c.copy_(a)
    with torch.cuda.stream(s1): d.copy_(b)
    torch.cuda.current_stream().wait_stream(s0); torch.cuda.current_stream().wait_stream(s1)
t(two, 16 * n, "2 x copy on two streams")
