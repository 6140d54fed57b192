"""In-tree builds of the native pieces (explicit hipcc / gcc; nothing is JIT-cached outside the repo)."""
import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
HDF5_ROOTS = [Path(os.environ.get("HDF5_ROOT", "/opt/conda")), Path("/usr"), Path("/usr/local")]


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(map(str, cmd)) + "\n" + r.stdout + r.stderr)
    return r.stdout + r.stderr


def _newer(target, sources):
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(s).stat().st_mtime > t for s in sources)


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and Path(c).exists():
            return c
    raise RuntimeError("hipcc not found")


def build_hip(force=False, verbose=False):
    """libpffdtd_hip.so: HIP kernels + C ABI, gfx950 only."""
    out = PKG / "libpffdtd_hip.so"
    srcs = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.inc")) + \
        [ROOT / "include" / "pffdtd_hip.h"]
    if force or _newer(out, srcs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-Wno-unused-value",  # numerics are stated per kernel with explicit fma where wanted
               "-I", str(ROOT / "include"), "-I", str(CSRC)] + \
            [str(s) for s in sorted(CSRC.glob("*.hip"))] + ["-o", str(out)]
        log = _run(cmd)
        if verbose:
            print(log)
    return out


def build_h5(force=False):
    """libpf_h5.so: HDF5 shim (gcc + system libhdf5)."""
    out = PKG / "libpf_h5.so"
    src = CSRC / "pf_h5.c"
    if not (force or _newer(out, [src])):
        return out
    for root in HDF5_ROOTS:
        inc, lib = root / "include", root / "lib"
        if (inc / "hdf5.h").exists() and any(lib.glob("libhdf5.so*")):
            _run(["gcc", "-O2", "-fPIC", "-shared", "-I", str(inc), str(src), "-o", str(out),
                  "-L", str(lib), "-lhdf5", f"-Wl,-rpath,{lib}"])
            return out
    raise RuntimeError("libhdf5 (hdf5.h + libhdf5.so) not found; set HDF5_ROOT")


def build_oracle(force=False):
    """oracle/libpf_oracle.so (test infrastructure) and, when /root/reference exists, oracle/_ref."""
    if force:
        _run(["make", "-C", str(ROOT / "oracle"), "clean"])
    _run(["make", "-C", str(ROOT / "oracle"), "oracle"])
    if Path("/root/reference/c_cuda/fdtd_main.c").exists():
        _run(["make", "-C", str(ROOT / "oracle"), "ref"])
    return ROOT / "oracle" / "libpf_oracle.so"


def build_all(force=False):
    return [build_hip(force), build_h5(force), build_oracle(force)]
