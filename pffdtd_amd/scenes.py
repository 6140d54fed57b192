"""The reference's four ready-made configurations (python/test_script_*.py) as keyword sets for `sim_setup`.

    test_script_CTK_cart_viz.py  -> ctk_cart_viz   BASELINE configs[0]: CTK church, 7-pt fp64, ~234x154x85, energy check
    test_script_CTK_cart_gpu.py  -> ctk_cart_gpu   BASELINE configs[1]: CTK church, 7-pt fp32, 894x579x309
    test_script_MV_fcc_gpu.py    -> mv_fcc_gpu     BASELINE configs[2]: Musikverein, 13-pt FCC fp32, 2852x1102x850 unfolded
    test_script_MV_fcc_viz.py    -> mv_fcc_viz     small FCC run for visualisation

The model exports and wall-impedance fits are data files of the reference (data/models, data/materials); copies
ship with the package (pffdtd_amd/data/models, pffdtd_amd/data/materials_DEF.npz; tests/golden/make_golden_materials.py made them).
"""
from pathlib import Path

import numpy as np

from . import h5io

CTK_MATS = {"AcousticPanel": "ctk_acoustic_panel.h5", "Altar": "ctk_altar.h5", "Carpet": "ctk_carpet.h5",
            "Ceiling": "ctk_ceiling.h5", "Glass": "ctk_window.h5", "PlushChair": "ctk_chair.h5", "Tile": "ctk_tile.h5",
            "Walls": "ctk_walls.h5"}
MV_MATS = {"Floor": "mv_floor.h5", "Chairs": "mv_chairs.h5", "Plasterboard": "mv_plasterboard.h5",
           "Window": "mv_window.h5", "Wood": "mv_wood.h5"}

CONFIGS = {
    "ctk_cart_viz": dict(model="CTK", mat_files_dict=CTK_MATS, source_num=1, insig_type="dhann30", diff_source=False,
                         duration=0.1, Tc=20, rh=50, fcc_flag=False, PPW=7.5, fmax=500.0),
    "ctk_cart_gpu": dict(model="CTK", mat_files_dict=CTK_MATS, source_num=1, insig_type="impulse", diff_source=True,
                         duration=3.0, Tc=20, rh=50, fcc_flag=False, PPW=10.5, fmax=1400.0),
    "mv_fcc_gpu": dict(model="MV", mat_files_dict=MV_MATS, source_num=3, insig_type="impulse", diff_source=True,
                       duration=3.0, Tc=20, rh=50, fcc_flag=True, PPW=7.7, fmax=2500.0),
    "mv_fcc_viz": dict(model="MV", mat_files_dict=MV_MATS, source_num=3, insig_type="dhann30", diff_source=False,
                       duration=0.1, Tc=20, rh=50, fcc_flag=True, PPW=5.6, fmax=1000.0),
}

DATA = Path(__file__).resolve().parent / "data"
MODEL_FILES = {"CTK": "CTK_Church_model_export.json", "MV": "MV_model_export.json.gz"}


def model_path(model, models_dir=None):
    return Path(models_dir or DATA / "models") / MODEL_FILES[model]


def write_materials(folder, npz=None):
    """Materialise the wall-impedance fits (DEF [Mb,3]) as <folder>/<name>.h5, the layout `SimMats.package` reads."""
    folder = Path(folder)
    folder.mkdir(parents=True, exist_ok=True)
    z = np.load(npz or DATA / "materials_DEF.npz")
    for name in z.files:
        h5io.write(folder / name, "DEF", z[name], append=False)
    return folder


def setup_kwargs(name, save_folder, mat_folder, models_dir=None, **override):
    """Keyword arguments for sim_setup() of a named configuration (override e.g. fmax/PPW/duration for small runs)."""
    cfg = dict(CONFIGS[name])
    model = cfg.pop("model")
    cfg.update(model_json_file=str(model_path(model, models_dir)), mat_folder=str(mat_folder), save_folder=str(save_folder))
    cfg.update(override)
    return cfg
