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


HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value"]
# numerics are stated per kernel with explicit fma where wanted, hence -ffp-contract=off
FILE_FLAGS = {"pf_tb2_fcc.hip": ["-fno-slp-vectorize"]}  # see the note at the top of that file


def _deps(src):
    """Headers a translation unit depends on (its #include "..." lines, followed transitively)."""
    seen, todo = set(), [src]
    while todo:
        f = todo.pop()
        for line in f.read_text().splitlines():
            line = line.strip()
            if line.startswith("#") and line[1:].lstrip().startswith('include "'):
                name = line.split('"')[1]
                for d in (CSRC, ROOT / "include"):
                    if (d / name).exists() and (d / name) not in seen:
                        seen.add(d / name)
                        todo.append(d / name)
    return [src] + sorted(seen)


def build_hip(force=False, verbose=False):
    """libpffdtd_hip.so: HIP kernels + C ABI, gfx950 only.  One object per .hip file (csrc/_obj/), relinked when any changed."""
    out = PKG / "libpffdtd_hip.so"
    objdir = CSRC / "_obj"
    objdir.mkdir(exist_ok=True)
    objs, relink = [], force or not out.exists()
    procs = []
    import hashlib
    for src in sorted(CSRC.glob("*.hip")):
        flags = [*HIP_FLAGS, *FILE_FLAGS.get(src.name, [])]
        if os.environ.get("PFFDTD_DEV_F32") == "1" and src.name == "pf_engine.hip":  # development only: fp32 engine alone, half the compile time
            flags.append("-DPF_DEV_F32_ONLY")
        # the flags are part of the object's name: a changed flag set (e.g. the -fno-slp-vectorize pf_tb2_fcc.hip needs for
        # bit-exactness) can never link a stale object
        tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:8]
        obj = objdir / f"{src.stem}.{tag}.o"
        objs.append(obj)
        if force or _newer(obj, _deps(src)):
            for old in objdir.glob(src.stem + ".*o"):
                if old != obj:
                    old.unlink()
            cmd = [hipcc(), *flags, "-c", "-I", str(ROOT / "include"), "-I", str(CSRC), str(src), "-o", str(obj)]
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            relink = True
    for cmd, p in procs:  # the translation units compile side by side
        log = p.communicate()[0]
        if p.returncode != 0:
            raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + log)
        if verbose:
            print(log)
    if relink or _newer(out, objs):
        _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [str(o) for o in objs] + ["-o", str(out)])
    return out


def build_probe(force=False):
    """tools/libpf_probe.so: calibration / research probes (tools/membench.py, tools/tb2_probe.py); not a product library."""
    tdir = ROOT / "tools"
    out = tdir / "libpf_probe.so"
    srcs = sorted((tdir / "csrc").glob("*")) + sorted(CSRC.glob("*.h"))
    if force or _newer(out, srcs):
        _run([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
              "-Wno-unused-value", "-I", str(tdir / "csrc"), "-I", str(CSRC), "-I", str(ROOT / "include"),
              str(tdir / "csrc" / "pf_probe.hip"), "-o", str(out)])
    return out


def load_probe():
    """ctypes handle of tools/libpf_probe.so with its two entry points typed."""
    import ctypes
    L = ctypes.CDLL(str(build_probe()))
    vp, i32, i64, d = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
    L.pf_probe_last_error.restype = ctypes.c_char_p
    L.pf_tb2_probe.restype = d
    L.pf_tb2_probe.argtypes = [vp, vp, vp, vp, i64, i64, i64, d, d, i32, i32, i32, i32]
    L.pf_membench.restype = d
    L.pf_membench.argtypes = [vp, vp, i64, i64, i64, i32, i32, i32, i32, i32, i32, i32]
    return L


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
        if (PKG / "libpffdtd_hip.so").exists():  # the reference's fdtd_main.c bound to the HIP library
            _run(["make", "-C", str(ROOT / "oracle"), "hipbind"])
    return ROOT / "oracle" / "libpf_oracle.so"


def build_all(force=False):
    return [build_hip(force), build_h5(force), build_oracle(force)]
