"""pffdtd_amd -- MI355X-native FDTD time-step engine behind bsxfun/pffdtd's engine seam and file contract.

Hot path: engine (ctypes binding of libpffdtd_hip.so, HIP only), sim_data (loader = load_sim_data mirror), h5io,
slab + dist (Z-slab multi-GPU), sim_fdtd / fdtd_main (drop-in CLIs).
Around it: room_geo + voxelizer (scene export -> boundary nodes, on
the device), setup_io + sim_setup + scenes (sim folders; the reference's test-script configurations), synth (synthetic
box scenes, rotate / fold / sort), process_outputs + air_abs (receivers -> room impulse responses).
"""
__version__ = "0.2.0"
