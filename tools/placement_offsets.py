#!/usr/bin/env python3
"""Round-5 experiment (verdict item: close the placement question): the four streams of k_tb3 carved from ONE allocation, with a
sweep of 2 MiB-multiple gaps between consecutive grids, against four separate allocations -- repeated over fresh allocations.
Prints ms per launch of k_tb3<float, 3, 8> (banded order, chunk 32) on a 1024^3 free-field grid for every layout.
usage: placement_offsets.py [n] [repeats]"""
import ctypes
import functools
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pffdtd_amd import build, engine  # noqa: E402

print = functools.partial(print, flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = build.load_probe()
vp, i32, i64, d = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
L.pf_tb3_probe.restype = d
L.pf_tb3_probe.argtypes = [vp, vp, vp, vp, i64, i64, i64, d, d, i32, i32, i32, i32]
P = engine.grid_pitch(n, 4)
gb = n * n * P * 4
MiB2 = 2 << 20


def time_layout(ptrs):
    return L.pf_tb3_probe(ptrs[0], ptrs[1], ptrs[2], ptrs[3], n, n, n, 1.0, 0.1, 8, 10308, 32, 4)


class _Loc(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("id", ctypes.c_int)]


class _Prop(ctypes.Structure):  # hipMemAllocationProp
    _fields_ = [("type", ctypes.c_int), ("requestedHandleType", ctypes.c_int), ("location", _Loc), ("win32HandleMetaData", ctypes.c_void_p),
                ("compressionType", ctypes.c_ubyte), ("gpuDirectRDMACapable", ctypes.c_ubyte), ("usage", ctypes.c_ushort)]


class _Acc(ctypes.Structure):  # hipMemAccessDesc
    _fields_ = [("location", _Loc), ("flags", ctypes.c_int)]


def one_physical_handle():
    """(c) the layout the round-4 verdict named: ALL four grids inside ONE hipMemCreate handle, mapped with hipMemMap -> ms per launch at
    gap 0 and with the roles interleaved, or None where the virtual-memory API is not available"""
    lib = next((ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln), None)  # (the HIP runtime torch loaded)
    if not lib:
        return None
    H = ctypes.CDLL(lib)
    for f in ("hipMemCreate", "hipMemAddressReserve", "hipMemMap", "hipMemSetAccess", "hipMemUnmap", "hipMemRelease", "hipMemAddressFree", "hipMemGetAllocationGranularity", "hipMemset"):
        if not hasattr(H, f):
            print("   (no", f, "in", lib + ")")
            return None
    prop = _Prop(1, 0, _Loc(1, torch.cuda.current_device()), None, 0, 0, 0)  # pinned device memory
    gran = ctypes.c_size_t(0)
    rc = H.hipMemGetAllocationGranularity(ctypes.byref(gran), ctypes.byref(prop), 1)  # recommended granularity
    if rc != 0 or gran.value == 0:
        print("   (hipMemGetAllocationGranularity:", rc, gran.value, ")")
        return None
    size = (4 * gb + gran.value - 1) // gran.value * gran.value
    handle, va = ctypes.c_void_p(), ctypes.c_void_p()
    rc = H.hipMemCreate(ctypes.byref(handle), ctypes.c_size_t(size), ctypes.byref(prop), ctypes.c_ulonglong(0))
    if rc != 0:
        print("   (hipMemCreate of", size >> 20, "MiB:", rc, ")")
        return None
    try:
        rc = H.hipMemAddressReserve(ctypes.byref(va), ctypes.c_size_t(size), ctypes.c_size_t(0), None, ctypes.c_ulonglong(0))
        if rc != 0:
            print("   (hipMemAddressReserve:", rc, ")")
            return None
        rc = H.hipMemMap(va, ctypes.c_size_t(size), ctypes.c_size_t(0), handle, ctypes.c_ulonglong(0))
        if rc != 0:
            print("   (hipMemMap:", rc, ")")
            return None
        acc = _Acc(_Loc(1, torch.cuda.current_device()), 3)
        rc = H.hipMemSetAccess(va, ctypes.c_size_t(size), ctypes.byref(acc), ctypes.c_size_t(1))
        if rc != 0:
            print("   (hipMemSetAccess:", rc, ")")
            return None
        if H.hipMemset(va, 0, ctypes.c_size_t(size)) != 0:
            print("   (hipMemset on the mapped range failed)")
            return None
        torch.cuda.synchronize()
        b = va.value
        t0 = time_layout([b + i * gb for i in range(4)])
        t1 = time_layout([b, b + 2 * gb, b + gb, b + 3 * gb])
        torch.cuda.synchronize()
        H.hipMemUnmap(va, ctypes.c_size_t(size))
        H.hipMemAddressFree(va, ctypes.c_size_t(size))
        return gran.value, t0, t1
    finally:
        H.hipMemRelease(handle)


for rep in range(reps):
    vmm = one_physical_handle()
    print(f"allocation {rep}: one hipMemCreate handle of 4 grids (granularity {vmm[0] >> 10} KiB), hipMemMap: gap 0 {vmm[1]:.3f} ms, roles interleaved {vmm[2]:.3f} ms" if vmm
          else f"allocation {rep}: hipMemCreate / hipMemMap not available")
    # (a) four separate allocations, as hipMalloc hands them out
    sep = [torch.zeros(gb // 4, dtype=torch.float32, device="cuda") for _ in range(4)]
    t_sep = time_layout([g.data_ptr() for g in sep])
    del sep
    torch.cuda.empty_cache()
    # (b) one allocation, grids at multiples of (grid + gap), gap = k x 2 MiB
    maxgap = 16
    big = torch.zeros((4 * (gb + maxgap * MiB2) + MiB2) // 4, dtype=torch.float32, device="cuda")
    base = (big.data_ptr() + MiB2 - 1) // MiB2 * MiB2
    row = []
    for k in range(maxgap + 1):
        stride = gb + k * MiB2
        row.append(time_layout([base + i * stride for i in range(4)]))
    # the same with the roles permuted (read grids 0, 2; write 1, 3) at gap 0
    t_perm = time_layout([base, base + 2 * gb, base + gb, base + 3 * gb])
    del big
    torch.cuda.empty_cache()
    print(f"allocation {rep}: separate {t_sep:.3f} ms | one allocation, gap k x 2 MiB, k = 0..{maxgap}: " + " ".join(f"{t:.3f}" for t in row)
          + f" | roles interleaved at gap 0: {t_perm:.3f}")
