"""ctypes wrapper of oracle/libpf_oracle.so -- TEST INFRASTRUCTURE (see pf_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        p = HERE / "libpf_oracle.so"
        if not p.exists():
            subprocess.run(["make", "-C", str(HERE), "oracle"], check=True, capture_output=True)
        L = ctypes.CDLL(str(p))
        for sfx in ("_f32", "_f64"):
            getattr(L, "oracle_run_sim" + sfx).restype = ctypes.c_double
            getattr(L, "oracle_run_sim" + sfx).argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                                           ctypes.POINTER(ctypes.c_double)]
            getattr(L, "oracle_run_sim_num" + sfx).restype = ctypes.c_double
            getattr(L, "oracle_run_sim_num" + sfx).argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                                               ctypes.POINTER(ctypes.c_double), ctypes.c_int]
            getattr(L, "orc_set_numerics" + sfx).argtypes = [ctypes.c_void_p, ctypes.c_int]
            getattr(L, "orc_set_slab_axis" + sfx).argtypes = [ctypes.c_void_p, ctypes.c_int]
            getattr(L, "orc_create" + sfx).restype = ctypes.c_void_p
            getattr(L, "orc_create" + sfx).argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
            getattr(L, "orc_step" + sfx).argtypes = [ctypes.c_void_p, ctypes.c_int64]
            getattr(L, "orc_destroy" + sfx).argtypes = [ctypes.c_void_p]
            getattr(L, "orc_grid" + sfx).restype = ctypes.c_void_p
            getattr(L, "orc_grid" + sfx).argtypes = [ctypes.c_void_p, ctypes.c_int]
        _LIB = L
        # OpenMP's default (one thread per visible CPU) is pathological under a cgroup CPU quota (the GPU box shows
        # 256 CPUs with a 16-CPU quota): spin-wait barriers of 256 threads on 16 CPUs take seconds per step.
        L.oracle_set_threads(default_threads())
    return _LIB


def default_threads():
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = Path("/sys/fs/cgroup/cpu.max").read_text().split()
    except (OSError, ValueError):
        q, p = "max", "1"
    if q != "max":
        n = max(1, min(n, int(int(q) / int(p))))
    return max(1, min(n, 64))


def _sfx(sd):
    return "_f32" if sd.real_bytes == 4 else "_f64"


def run_sim(sd, threads=None, safeguarded=False):
    """oracle analogue of run_sim(struct SimData*): fills sd.u_out, returns (elapsed, t_air, t_bn).
    safeguarded=True: the arithmetic of the reference's CUDA engine instead of its C CPU engine's (parity unpinned)."""
    L = lib()
    if threads:
        L.oracle_set_threads(int(threads))
    s = sd.as_struct()
    ta, tb = ctypes.c_double(), ctypes.c_double()
    el = getattr(L, "oracle_run_sim_num" + _sfx(sd))(ctypes.byref(s), ctypes.byref(ta), ctypes.byref(tb), int(bool(safeguarded)))
    if el < 0:
        raise RuntimeError("oracle_run_sim failed")
    return el, ta.value, tb.value


class Engine:
    """Step-wise oracle engine (used by the slab tests as the per-slab stepper and for grid comparisons)."""

    def __init__(self, sd, slab_first=True, slab_last=True, safeguarded=False, along_z=False):
        if getattr(sd, "bn_mask", None) is None:
            raise ValueError("the oracle steps by the boundary-node bit mask (cpu_engine.h:175-194): build the SimData with build_mask=True")
        self.L = lib()
        self.sd = sd
        self.sfx = _sfx(sd)
        self._s = sd.as_struct()
        self.h = getattr(self.L, "orc_create" + self.sfx)(ctypes.byref(self._s), int(slab_first), int(slab_last))
        if not self.h:
            raise RuntimeError("orc_create failed")
        if safeguarded:
            getattr(self.L, "orc_set_numerics" + self.sfx)(self.h, 1)
        if along_z:  # a slab of a chain cut along file z: the flags gate the z flips
            getattr(self.L, "orc_set_slab_axis" + self.sfx)(self.h, 1)

    def step(self, n):
        getattr(self.L, "orc_step" + self.sfx)(self.h, int(n))

    def grid(self, which):
        """numpy VIEW of u0 (which=0) / u1 (which=1), shape (Nx,Ny,Nz); re-fetch after each step (pointers rotate)."""
        p = getattr(self.L, "orc_grid" + self.sfx)(self.h, int(which))
        n = self.sd.Npts
        ct = ctypes.c_float if self.sfx == "_f32" else ctypes.c_double
        a = np.ctypeslib.as_array((ct * n).from_address(p))
        return a.reshape(self.sd.Nx, self.sd.Ny, self.sd.Nz)

    def close(self):
        if self.h:
            getattr(self.L, "orc_destroy" + self.sfx)(self.h)
            self.h = None

    def __del__(self):
        self.close()


def ref_binary(precision):
    p = HERE / "_ref" / f"fdtd_main_cpu_{'single' if precision in ('single', 1) else 'double'}.x"
    return p if p.exists() else None


def run_reference(data_dir, precision, threads=None):
    """Run the compiled reference (oracle/_ref) in data_dir; returns u_out as written to sim_outs.h5."""
    import os
    from pffdtd_amd import h5io
    exe = ref_binary(precision)
    if exe is None:
        raise FileNotFoundError("oracle/_ref not built (needs /root/reference)")
    env = dict(os.environ)
    if threads:
        env["OMP_NUM_THREADS"] = str(threads)
    env["TERM"] = env.get("TERM", "dumb")
    r = subprocess.run([str(exe)], cwd=str(data_dir), capture_output=True, text=True, env=env)
    if r.returncode != 0:
        raise RuntimeError(f"reference binary failed ({r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
    return h5io.read(Path(data_dir) / "sim_outs.h5", "u_out"), r.stdout
