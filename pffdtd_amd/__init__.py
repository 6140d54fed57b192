"""pffdtd_amd -- MI355X-native FDTD time-step engine behind bsxfun/pffdtd's engine seam and file contract.

Modules: engine (ctypes binding of libpffdtd_hip.so, HIP only), sim_data (loader = load_sim_data mirror), h5io,
synth (synthetic scenes + rotate/fold/sort), slab + dist (Z-slab multi-GPU), sim_fdtd / fdtd_main (drop-in CLIs),
setup_io (SimConsts / SimComms / SimMats writers), process_outputs (receiver post-processing).
"""
__version__ = "0.1.0"
