"""ctypes binding of libpffdtd_hip.so (include/pffdtd_hip.h): the only compute path of this package.

There is no CPU fallback: if the HIP library is missing or no GPU is visible, construction raises.
"""
import ctypes
import importlib.util
import os
import sys
from pathlib import Path

import numpy as np

from .sim_data import PfSimData

_HERE = Path(__file__).resolve().parent
_LIB = None

PF_NUM_CPU_EXACT = 0
PF_NUM_GPU_SAFEGUARDED = 2


class PfOpts(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("numerics", ctypes.c_int32), ("slab_first", ctypes.c_int32),
                ("slab_last", ctypes.c_int32), ("readout_chunk", ctypes.c_int32), ("air_variant", ctypes.c_int32),
                ("air_chunk", ctypes.c_int32), ("timing", ctypes.c_int32), ("ext_u0", ctypes.c_void_p),
                ("ext_u1", ctypes.c_void_p), ("x_global0", ctypes.c_int32), ("layout", ctypes.c_int32),
                ("energy", ctypes.c_int32), ("multi_flags", ctypes.c_int32), ("transport", ctypes.c_int32),
                ("verify_exchange", ctypes.c_int32), ("only_slab", ctypes.c_int32), ("wall_scale", ctypes.c_double)]


# Development / test switches (csrc/pf_debug.h: the PF_DBG_* bits, the chain's fault injection).  Not fields of pf_opts: they reach the
# library through its internal hook, for the NEXT create call of this thread.
_HOOKS = ("debug", "test_drop_exchange", "test_faults")


def _set_hooks(debug=0, test_drop_exchange=0, test_faults=0):
    lib().pf_internal_hooks(int(debug), int(test_drop_exchange), int(test_faults))


class PfTiming(ctypes.Structure):
    _fields_ = [("air_ms_total", ctypes.c_double), ("air_launches", ctypes.c_int64),
                ("step_ms_total", ctypes.c_double), ("steps", ctypes.c_int64),
                ("tb2_ms_total", ctypes.c_double), ("tb2_launches", ctypes.c_int64), ("tb2_cells", ctypes.c_int64),
                ("tune_ms", ctypes.c_double * 3), ("air_path", ctypes.c_int64), ("tb2_lw", ctypes.c_int64),
                ("tb2_dirty_tiles", ctypes.c_int64), ("place_candidates", ctypes.c_int64), ("place_ms", ctypes.c_double * 3),
                ("wall_blocks", ctypes.c_int64 * 2), ("tb_steps_per_pass", ctypes.c_int64), ("wall_bricks", ctypes.c_int64), ("wall_three_steps", ctypes.c_int64)]


class PfMultiInfo(ctypes.Structure):
    _fields_ = [("nslabs", ctypes.c_int32), ("transport", ctypes.c_int32), ("rccl_self", ctypes.c_int32),
                ("exchange_verified", ctypes.c_int32), ("exchanges_checked", ctypes.c_int64),
                ("exchange_nonzero", ctypes.c_int32), ("cut_along_z", ctypes.c_int32), ("plane_bytes", ctypes.c_int64),
                ("last_run_seconds", ctypes.c_double), ("transport_name", ctypes.c_char * 64),
                ("transport_note", ctypes.c_char * 256), ("wall_scale", ctypes.c_double), ("wall_measured", ctypes.c_int32)]


class PfError(RuntimeError):
    pass


EXPORTS = ["pf_last_error", "pf_version", "pf_device_count", "pf_grid_bytes", "pf_grid_pitch", "pf_opts_default",
           "pf_run_sim", "pf_engine_create", "pf_engine_destroy", "pf_engine_run", "pf_engine_step_begin",
           "pf_engine_halo_ptrs", "pf_engine_step_end", "pf_engine_state_grids", "pf_engine_layout", "pf_engine_place_grids", "pf_engine_place_grids5", "pf_engine_set_spares", "pf_engine_stream", "pf_engine_sync",
           "pf_engine_flush_outputs", "pf_engine_get_grid", "pf_engine_set_grid", "pf_engine_timing", "pf_engine_set_timing",
           "pf_engine_energy_cfg", "pf_engine_run_energy", "pf_run_sim_devices", "pf_slab_partition", "pf_slab_partition_w", "pf_slab_partition_axis", "pf_slab_wall_scale",
           "pf_multi_create", "pf_multi_run", "pf_multi_get_info", "pf_multi_get_slab", "pf_multi_destroy",
           "pf_rccl_unique_id", "pf_rccl_comm_create", "pf_rccl_exchange", "pf_rccl_comm_destroy"]


INTERNAL_EXPORTS = ["pf_internal_hooks"]  # csrc/pf_debug.h: exported, but no part of the drop-in boundary (not in include/)

PF_MULTI_EVEN_SPLIT, PF_MULTI_ONE_THREAD, PF_MULTI_NO_PAIRS, PF_MULTI_FORCE_PAIRS, PF_MULTI_CUT_Z, PF_MULTI_CUT_X = 1, 2, 4, 8, 16, 32
PF_MULTI_NO_TRIPLES = 64
PF_MULTI_MEASURE_WEIGHTS = 128
PF_TRANSPORT_AUTO, PF_TRANSPORT_PEER, PF_TRANSPORT_RCCL, PF_TRANSPORT_HOST = 0, 1, 2, 3
PF_LAYOUT_AUTO, PF_LAYOUT_EXCHANGED, PF_LAYOUT_FILE = 0, 1, 2


def _preload_torch_hip():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same soname as /opt/rocm's).  The copy loaded first serves
    the whole process, and torch only finds its GPUs through its own.  So that the order of `import torch` and the first
    engine call does not matter, load torch's copy (without importing torch) before libpffdtd_hip.so pulls in the system
    one.  PFFDTD_SYSTEM_HIP=1 keeps the system runtime (engine-only hosts that never import torch)."""
    if "torch" in sys.modules or os.environ.get("PFFDTD_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    p = Path(spec.submodule_search_locations[0]) / "lib" / "libamdhip64.so"
    if p.exists():
        try:
            ctypes.CDLL(str(p), mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib_path():
    return _HERE / "libpffdtd_hip.so"


def lib():
    """Load libpffdtd_hip.so; raises if it has not been built (no silent fallback)."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not p.exists():
            raise PfError(f"{p} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _preload_torch_hip()
        L = ctypes.CDLL(str(p))
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
        L.pf_last_error.restype = ctypes.c_char_p
        L.pf_version.restype = ctypes.c_char_p
        L.pf_device_count.restype = ctypes.c_int
        L.pf_grid_bytes.restype = ctypes.c_size_t
        L.pf_grid_bytes.argtypes = [i64, i64, i64, i32]
        L.pf_grid_pitch.restype = i64
        L.pf_grid_pitch.argtypes = [i64, i32]
        L.pf_opts_default.argtypes = [ctypes.POINTER(PfOpts)]
        L.pf_internal_hooks.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        L.pf_internal_hooks.restype = None
        L.pf_run_sim.restype = ctypes.c_double
        L.pf_run_sim.argtypes = [ctypes.POINTER(PfSimData)]
        L.pf_engine_create.argtypes = [ctypes.POINTER(PfSimData), ctypes.POINTER(PfOpts), ctypes.POINTER(vp)]
        L.pf_engine_destroy.argtypes = [vp]
        L.pf_engine_destroy.restype = None
        L.pf_engine_run.argtypes = [vp, i64, i64]
        L.pf_engine_step_begin.argtypes = [vp, i64]
        L.pf_engine_step_end.argtypes = [vp, i64]
        L.pf_engine_halo_ptrs.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                          ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
        L.pf_engine_state_grids.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
        L.pf_engine_layout.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i32)]
        L.pf_engine_place_grids.argtypes = [vp, ctypes.POINTER(vp), i32, ctypes.POINTER(i32)]
        L.pf_engine_place_grids5.argtypes = [vp, ctypes.POINTER(vp), i32, ctypes.POINTER(i32)]
        L.pf_engine_stream.restype = vp
        L.pf_engine_stream.argtypes = [vp, i32]
        L.pf_engine_sync.argtypes = [vp]
        L.pf_engine_flush_outputs.argtypes = [vp]
        L.pf_engine_get_grid.argtypes = [vp, i32, vp]
        L.pf_engine_set_grid.argtypes = [vp, i32, vp]
        L.pf_engine_timing.argtypes = [vp, ctypes.POINTER(PfTiming), i32]
        L.pf_engine_set_timing.argtypes = [vp, i32]
        dp = ctypes.POINTER(ctypes.c_double)
        L.pf_engine_energy_cfg.argtypes = [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, dp]
        L.pf_engine_run_energy.argtypes = [vp, i64, i64, dp, dp, dp]
        L.pf_run_sim_devices.restype = ctypes.c_double
        L.pf_run_sim_devices.argtypes = [ctypes.POINTER(PfSimData), i32, ctypes.POINTER(i32), ctypes.POINTER(PfOpts)]
        L.pf_slab_partition.argtypes = [ctypes.POINTER(PfSimData), i32, i32, ctypes.POINTER(i64)]
        L.pf_slab_partition_w.argtypes = [ctypes.POINTER(PfSimData), i32, i32, ctypes.c_double, ctypes.POINTER(i64)]
        L.pf_slab_partition_axis.argtypes = [ctypes.POINTER(PfSimData), i32, i32, ctypes.c_double, i32, ctypes.POINTER(i64)]
        L.pf_slab_wall_scale.argtypes = [ctypes.POINTER(PfSimData), i32, i32, ctypes.POINTER(PfOpts)]
        L.pf_slab_wall_scale.restype = ctypes.c_double
        L.pf_engine_set_spares.argtypes = [vp, vp, vp]
        L.pf_multi_create.argtypes = [ctypes.POINTER(PfSimData), i32, ctypes.POINTER(i32), ctypes.POINTER(PfOpts), ctypes.POINTER(vp)]
        L.pf_multi_run.argtypes = [vp, i64, i64]
        L.pf_multi_get_info.argtypes = [vp, ctypes.POINTER(PfMultiInfo)]
        L.pf_multi_get_slab.argtypes = [vp, i32, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(vp)]
        L.pf_multi_destroy.argtypes = [vp]
        L.pf_multi_destroy.restype = None
        L.pf_rccl_unique_id.argtypes = [ctypes.c_char_p]
        L.pf_rccl_comm_create.argtypes = [ctypes.c_char_p, i32, i32, i32, ctypes.POINTER(ctypes.c_void_p)]
        L.pf_rccl_exchange.argtypes = [ctypes.c_void_p, ctypes.c_void_p, i32, i32]
        L.pf_rccl_comm_destroy.argtypes = [ctypes.c_void_p]
        L.pf_rccl_comm_destroy.restype = None
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise PfError(f"pffdtd_hip error {rc}: {lib().pf_last_error().decode()}")


def device_count():
    return int(lib().pf_device_count())


def grid_pitch(Nz, real_bytes):
    return int(lib().pf_grid_pitch(int(Nz), int(real_bytes)))


def run_sim(sd):
    """`double run_sim(struct SimData*)` (cpu_engine.h:52 / gpu_engine.h:665): fills sd.u_out, returns seconds."""
    s = sd.as_struct()
    el = lib().pf_run_sim(ctypes.byref(s))
    if el < 0:
        raise PfError(f"pf_run_sim failed: {lib().pf_last_error().decode()}")
    return el


def run_sim_devices(sd, devices, multi_flags=0, **opts):
    """pf_run_sim_devices: run_sim on a chain of Z-slabs, slab g on HIP device devices[g] (ids may repeat = virtual
    slabs on one GPU).  opts: numerics, air_variant, readout_chunk, debug.  Fills sd.u_out, returns seconds."""
    L = lib()
    s = sd.as_struct()
    o = _multi_opts(multi_flags, opts)
    devs = (ctypes.c_int32 * len(devices))(*[int(d) for d in devices])
    el = L.pf_run_sim_devices(ctypes.byref(s), len(devices), devs, ctypes.byref(o))
    if el < 0:
        raise PfError(f"pf_run_sim_devices failed: {L.pf_last_error().decode()}")
    return el


def _multi_opts(multi_flags, opts):
    """pf_opts from keyword arguments; `debug` / `test_*` go to the library's internal hook (pending for the create call that follows)."""
    o = PfOpts()
    lib().pf_opts_default(ctypes.byref(o))
    hooks = {k: opts[k] for k in _HOOKS if k in opts}
    for k, v in opts.items():
        if k not in _HOOKS:
            setattr(o, k, float(v) if k == "wall_scale" else int(v))
    o.multi_flags = int(multi_flags)
    _set_hooks(**hooks)
    return o


class HipMulti:
    """A chain of Z-slabs on several devices behind ONE C object (pf_multi_create): one persistent host thread per slab,
    ghost planes by peer copies or RCCL.  devices[g] = HIP device of slab g (ids may repeat: virtual slabs on one GPU).
    opts: pf_opts fields common to all slabs (numerics, air_variant, readout_chunk, debug, timing, transport,
    verify_exchange)."""

    def __init__(self, sd, devices, multi_flags=0, **opts):
        L = lib()
        self.sd = sd
        self._s = sd.as_struct()
        o = _multi_opts(multi_flags, opts)
        devs = (ctypes.c_int32 * len(devices))(*[int(d) for d in devices])
        self._h = ctypes.c_void_p()
        _check(L.pf_multi_create(ctypes.byref(self._s), len(devices), devs, ctypes.byref(o), ctypes.byref(self._h)))
        self.nslabs = len(devices)

    def run(self, n0, nsteps):
        _check(lib().pf_multi_run(self._h, int(n0), int(nsteps)))

    def info(self):
        i = PfMultiInfo()
        _check(lib().pf_multi_get_info(self._h, ctypes.byref(i)))
        return {"nslabs": i.nslabs, "transport": i.transport, "transport_name": i.transport_name.decode(),
                "rccl_self": bool(i.rccl_self), "exchanges_checked": i.exchanges_checked,
                "exchange_verified": None if i.exchange_verified < 0 else bool(i.exchange_verified),
                "exchange_nonzero": bool(i.exchange_nonzero), "cut_along_z": bool(i.cut_along_z), "plane_bytes": i.plane_bytes,
                "last_run_seconds": i.last_run_seconds, "transport_note": i.transport_note.decode(),
                "wall_scale": i.wall_scale, "wall_measured": bool(i.wall_measured)}

    def slab(self, g):
        """-> dict(x0, x1, device, paired, steps_per_pass, engine): engine = a non-owning HipEngine view of slab g's engine (state_grids, timing)"""
        x0, x1, dev, pr, eh = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_void_p()
        _check(lib().pf_multi_get_slab(self._h, int(g), ctypes.byref(x0), ctypes.byref(x1), ctypes.byref(dev), ctypes.byref(pr), ctypes.byref(eh)))
        # (the C chain reports 0 = single steps, 1 = pairs, 3 = triples: here `paired` is strictly a boolean, `steps_per_pass` 0 / 2 / 3)
        spp = {0: 0, 1: 2, 3: 3}.get(int(pr.value), 2)
        return {"x0": x0.value, "x1": x1.value, "device": dev.value, "paired": spp > 0, "steps_per_pass": spp, "engine": _EngineView(eh, self.sd.real_bytes)}

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().pf_multi_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _EngineView:
    """Borrowed handle of an engine owned by a HipMulti: inspection calls only."""

    def __init__(self, handle, real_bytes):
        self._h = handle
        self.dtype = np.float32 if real_bytes == 4 else np.float64

    state_grids = lambda self: HipEngine.state_grids(self)  # noqa: E731
    layout = lambda self: HipEngine.layout(self)  # noqa: E731
    timing = lambda self, reset=False: HipEngine.timing(self, reset)  # noqa: E731
    sync = lambda self: HipEngine.sync(self)  # noqa: E731
    set_timing = lambda self, on: HipEngine.set_timing(self, on)  # noqa: E731


def slab_partition(sd, nslabs, even=False, wall_scale=1.0, along_z=False):
    """Owned plane ranges [(x0, x1)] of pf_run_sim_devices' slabs (wall_scale: factor on the wall planes' weights; along_z: the cut
    along FILE Z of rooms stored with exchanged axes).  Host code only: works without a device."""
    s = sd.as_struct()
    cuts = (ctypes.c_int64 * (nslabs + 1))()
    _check(lib().pf_slab_partition_axis(ctypes.byref(s), int(nslabs), int(bool(even)), float(wall_scale), int(bool(along_z)), cuts))
    return [(int(cuts[g]), int(cuts[g + 1])) for g in range(nslabs)]


def slab_wall_scale(sd, nslabs, device=0, **opts):
    """pf_slab_wall_scale: the factor on the wall planes' weights measured for this scene on `device` (None: not measured)."""
    s = sd.as_struct()
    o = _multi_opts(0, opts)
    k = float(lib().pf_slab_wall_scale(ctypes.byref(s), int(nslabs), int(device), ctypes.byref(o)))
    return k if k > 0 else None


class HipEngine:
    """One engine instance = one grid (or one Z-slab) resident on one MI355X."""

    def __init__(self, sd, device=0, numerics=PF_NUM_CPU_EXACT, slab_first=True, slab_last=True, air_variant=0,
                 air_chunk=0, timing=False, readout_chunk=0, ext_u0=None, ext_u1=None, debug=0, x_global0=0, energy=False, layout=0):
        L = lib()
        self.sd = sd
        self._s = sd.as_struct()
        o = PfOpts()
        L.pf_opts_default(ctypes.byref(o))
        o.device, o.numerics = int(device), int(numerics)
        o.slab_first, o.slab_last = int(bool(slab_first)), int(bool(slab_last))
        o.air_variant, o.air_chunk, o.timing, o.readout_chunk = int(air_variant), int(air_chunk), int(bool(timing)), \
            int(readout_chunk)
        o.x_global0 = int(x_global0)
        o.layout = int(layout)
        o.energy = int(bool(energy))
        if ext_u0 is not None and ext_u1 is not None:
            o.ext_u0, o.ext_u1 = int(ext_u0), int(ext_u1)
        self._h = ctypes.c_void_p()
        _set_hooks(debug=debug)
        _check(L.pf_engine_create(ctypes.byref(self._s), ctypes.byref(o), ctypes.byref(self._h)))
        self.dtype = np.float32 if sd.real_bytes == 4 else np.float64

    def run(self, n0, nsteps):
        _check(lib().pf_engine_run(self._h, int(n0), int(nsteps)))

    def step_begin(self, n):
        _check(lib().pf_engine_step_begin(self._h, int(n)))

    def step_end(self, n):
        _check(lib().pf_engine_step_end(self._h, int(n)))

    def halo_ptrs(self):
        vp = ctypes.c_void_p
        slo, shi, rlo, rhi, nb = vp(), vp(), vp(), vp(), ctypes.c_size_t()
        _check(lib().pf_engine_halo_ptrs(self._h, ctypes.byref(slo), ctypes.byref(shi), ctypes.byref(rlo),
                                         ctypes.byref(rhi), ctypes.byref(nb)))
        return slo.value, shi.value, rlo.value, rhi.value, nb.value

    def state_grids(self):
        """Device pointers (u^{n-1}, u^n) of the state grids between runs (pf_engine_state_grids)."""
        vp = ctypes.c_void_p
        up, uc = vp(), vp()
        _check(lib().pf_engine_state_grids(self._h, ctypes.byref(up), ctypes.byref(uc)))
        return up.value, uc.value

    def layout(self):
        """-> ((planes, rows, columns) as stored, pitch in elements, exchanged): exchanged = the engine stores the file's x and z
        axes exchanged; a state grid holds planes * rows * pitch elements (pf_engine_layout)."""
        dims, pitch, ex = (ctypes.c_int64 * 3)(), ctypes.c_int64(), ctypes.c_int32()
        _check(lib().pf_engine_layout(self._h, dims, ctypes.byref(pitch), ctypes.byref(ex)))
        return (int(dims[0]), int(dims[1]), int(dims[2])), int(pitch.value), bool(ex.value)

    def place_grids(self, ptrs):
        """Offer a pool of >= 4 zero-filled caller-owned grids to a slab engine (pf_engine_place_grids).  -> (paired, idx):
        idx[0:2] = positions of the state grids in the pool, idx[2:4] = the spares (or -1 when it steps singly)."""
        arr = (ctypes.c_void_p * len(ptrs))(*[int(p) for p in ptrs])
        idx = (ctypes.c_int32 * 4)()
        _check(lib().pf_engine_place_grids(self._h, arr, len(ptrs), idx))
        return idx[2] >= 0, list(idx)

    def place_grids5(self, ptrs):
        """The same with room for triples (pf_engine_place_grids5).  -> (steps per pass: 3 triples, 2 pairs, 0 single steps; idx[0:5]):
        idx[2:4] = the grids the blocked kernel writes, idx[4] = the u^{n+1} grid of the triples (or -1)."""
        arr = (ctypes.c_void_p * len(ptrs))(*[int(p) for p in ptrs])
        idx = (ctypes.c_int32 * 5)()
        _check(lib().pf_engine_place_grids5(self._h, arr, len(ptrs), idx))
        return (3 if idx[4] >= 0 else (2 if idx[2] >= 0 else 0)), list(idx)

    def set_spares(self, ptr2, ptr3):
        """Two more caller-owned state grids: lets a slab engine step in temporally blocked pairs.  -> True if it will."""
        rc = lib().pf_engine_set_spares(self._h, ctypes.c_void_p(ptr2), ctypes.c_void_p(ptr3))
        if rc not in (0, 1):
            _check(rc)
        return rc == 0

    def stream(self, which):
        return lib().pf_engine_stream(self._h, int(which))

    def sync(self):
        _check(lib().pf_engine_sync(self._h))

    def flush_outputs(self):
        _check(lib().pf_engine_flush_outputs(self._h))

    def get_grid(self, which):
        a = np.empty((self.sd.Nx, self.sd.Ny, self.sd.Nz), dtype=self.dtype)
        _check(lib().pf_engine_get_grid(self._h, int(which), a.ctypes.data_as(ctypes.c_void_p)))
        return a

    def set_grid(self, which, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.size == self.sd.Npts
        _check(lib().pf_engine_set_grid(self._h, int(which), a.ctypes.data_as(ctypes.c_void_p)))

    def energy_cfg(self, h, c, Ts, DEF_list):
        """DEF_list: the materials' (Mb,3) arrays (sim_mats.h5 mat_XX_DEF)."""
        tab = np.zeros((max(len(DEF_list), 1), 12, 3), dtype=np.float64)
        for k, d in enumerate(DEF_list):
            tab[k, :d.shape[0]] = d
        self._DEF = tab
        dp = ctypes.POINTER(ctypes.c_double)
        _check(lib().pf_engine_energy_cfg(self._h, float(h), float(c), float(Ts), tab.ctypes.data_as(dp)))

    def run_energy(self, n0, nsteps, H_tot, E_lost, E_in):
        dp = ctypes.POINTER(ctypes.c_double)
        for a in (H_tot, E_lost, E_in):
            assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
        _check(lib().pf_engine_run_energy(self._h, int(n0), int(nsteps), H_tot.ctypes.data_as(dp),
                                          E_lost.ctypes.data_as(dp), E_in.ctypes.data_as(dp)))

    def timing(self, reset=False):
        t = PfTiming()
        _check(lib().pf_engine_timing(self._h, ctypes.byref(t), int(reset)))
        return {"air_ms_total": t.air_ms_total, "air_launches": t.air_launches, "step_ms_total": t.step_ms_total,
                "steps": t.steps, "tb2_ms_total": t.tb2_ms_total, "tb2_launches": t.tb2_launches, "tb2_cells": t.tb2_cells,
                "tune_ms": list(t.tune_ms), "air_path": t.air_path, "tb2_lw": t.tb2_lw, "tb2_dirty_tiles": t.tb2_dirty_tiles,
                "place_candidates": t.place_candidates, "place_ms": list(t.place_ms), "wall_blocks": list(t.wall_blocks),
                "tb_steps_per_pass": t.tb_steps_per_pass, "wall_bricks": t.wall_bricks, "wall_three_steps": t.wall_three_steps}

    def set_timing(self, on):
        _check(lib().pf_engine_set_timing(self._h, int(bool(on))))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().pf_engine_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
