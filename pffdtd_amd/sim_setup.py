"""Set up a simulation folder from a scene export: grid, sources/receivers, materials, voxelization, GPU-prep.

Same call signature and the same sequence as the reference's `sim_setup()` (python/sim_setup.py:29-146): RoomGeo ->
SimConsts -> SimMats -> CartGrid -> SimComms -> voxelize -> clash check -> (copy, rotate, fold, sort) for the GPU
folder.  The reference's `VoxGrid` stage and its process pool have no counterpart (the voxelizer bins on the
device); `draw_vox`, `draw_backend`, `Nvox_est`, `Nh`, `Nprocs` are accepted and ignored.  Writes sim_consts.h5,
sim_mats.h5, cart_grid.h5, comms_out.h5, vox_out.h5 with the reference's dataset names and dtypes.
"""
import shutil
from pathlib import Path

import numpy as np

from . import setup_io
from .room_geo import RoomGeo
from .voxelizer import VoxScene


def sim_setup(insig_type=None, fmax=None, PPW=None, save_folder=None, model_json_file=None, mat_folder=None,
              mat_files_dict=None, duration=None, Tc=20, rh=50, source_num=1, save_folder_gpu=None, draw_vox=False,
              draw_backend=None, diff_source=False, fcc_flag=False, bmin=None, bmax=None, Nvox_est=None, Nh=None,
              Nprocs=None, compress=None, rot_az_el=(0.0, 0.0), device=0, check_adj=True):
    for name, v in (("insig_type", insig_type), ("fmax", fmax), ("PPW", PPW), ("save_folder", save_folder),
                    ("model_json_file", model_json_file), ("mat_folder", mat_folder), ("mat_files_dict", mat_files_dict),
                    ("duration", duration)):
        if v is None:
            raise ValueError(f"sim_setup: {name} is required")
    if source_num < 1:
        raise ValueError("source_num is one-based")
    if draw_vox:
        raise ValueError("drawing is out of scope")
    if bmin is not None and bmax is not None:  # custom scene bounds (open scenes)
        bmin, bmax = np.array(bmin, dtype=np.float64), np.array(bmax, dtype=np.float64)

    room_geo = RoomGeo(model_json_file, az_el=rot_az_el, bmin=bmin, bmax=bmax)
    room_geo.print_stats()
    Sxyz = room_geo.Sxyz[source_num - 1]  # one source, many receivers
    Rxyz = room_geo.Rxyz

    sim_consts = setup_io.SimConsts(Tc=Tc, rh=rh, fmax=fmax, PPW=PPW, fcc=fcc_flag)
    sim_consts.save(save_folder)

    sim_mats = setup_io.SimMats(save_folder=save_folder)
    sim_mats.package(mat_files_dict=mat_files_dict, mat_list=room_geo.mat_str, read_folder=mat_folder)

    cart_grid = setup_io.CartGrid(h=sim_consts.h, offset=3.5, bmin=room_geo.bmin, bmax=room_geo.bmax, fcc=fcc_flag)
    cart_grid.save(save_folder)

    sim_comms = setup_io.SimComms(save_folder=save_folder)  # reads the two files just written, like the reference
    sim_comms.prepare_source_pts(Sxyz)
    sim_comms.prepare_receiver_pts(Rxyz)
    sim_comms.prepare_source_signals(duration, sig_type=insig_type)
    if diff_source:
        sim_comms.diff_source()
    sim_comms.save(compress=compress)

    vox_scene = VoxScene(room_geo, cart_grid, fcc=fcc_flag, device=device)
    vox_scene.calc_adj()
    if check_adj:
        # The reference calls check_adj_full() here too, but its asserts cannot fire (see voxelizer.check_adj_full), and
        # its voxelization of its own CTK model at the test-script resolution does leave a handful of one-sided legs
        # (8 of 2.3e7).  The output is kept identical to the reference's; check_adj="strict" turns the count into an error.
        bad = vox_scene.check_adj_full()
        if bad and check_adj == "strict":
            raise RuntimeError(f"voxelization left {bad} legs cut from one end only (stability precondition)")
        if bad:
            print(f"--SIM_SETUP: warning: {bad} legs are cut from one end only (same as the reference voxelizer)")
    vox_scene.save(save_folder, compress=compress)
    sim_comms.check_for_clashes(vox_scene.bn_ixyz)

    if save_folder_gpu is not None:
        if Path(save_folder_gpu) != Path(save_folder):
            Path(save_folder_gpu).mkdir(parents=True, exist_ok=True)
            for f in Path(save_folder).glob("*.h5"):  # copy_sim_data (rotate_sim_data.py:264-276)
                shutil.copyfile(f, Path(save_folder_gpu) / f.name)
        setup_io.prep_folder(save_folder_gpu, rotate=True, fold=bool(fcc_flag), sort=True, compress=int(compress or 0))
    return vox_scene


def main():
    """python -m pffdtd_amd.sim_setup --config ctk_cart_gpu --save_folder DIR [--save_folder_gpu DIR2] [--mat_folder M]
    [--fmax F --PPW P --duration T]: one of the reference's test-script configurations (pffdtd_amd/scenes.py), from the
    scene exports and wall-impedance fits that travel with the tests."""
    import argparse
    from . import scenes
    p = argparse.ArgumentParser()
    p.add_argument("--config", required=True, choices=sorted(scenes.CONFIGS))
    p.add_argument("--save_folder", required=True)
    p.add_argument("--save_folder_gpu", default=None)
    p.add_argument("--mat_folder", default=None, help="folder with the material .h5 files (default: written from the fixtures)")
    p.add_argument("--fmax", type=float, default=None)
    p.add_argument("--PPW", type=float, default=None)
    p.add_argument("--duration", type=float, default=None)
    p.add_argument("--gpu", type=int, default=0)
    a = p.parse_args()
    mats = a.mat_folder or scenes.write_materials(Path(a.save_folder) / "materials")
    over = {k: v for k, v in (("fmax", a.fmax), ("PPW", a.PPW), ("duration", a.duration)) if v is not None}
    sim_setup(**scenes.setup_kwargs(a.config, a.save_folder, mats, save_folder_gpu=a.save_folder_gpu, compress=0, device=a.gpu, **over))


if __name__ == "__main__":
    main()
