#!/usr/bin/env python3
"""Input fixtures for the end-to-end setup tests, taken from data files the reference ships for its own test
scripts (no source code): the fitted wall-impedance branches data/materials/*.h5 (DEF arrays, [Mb,3] doubles)
-> materials_DEF.npz, and the Musikverein scene export (1.7 MB of JSON) -> models/MV_model_export.json.gz."""
import gzip
import shutil
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
DATA = HERE.parent.parent / "pffdtd_amd" / "data"
sys.path.insert(0, str(HERE.parent.parent))
from pffdtd_amd import h5io  # noqa: E402

REF = Path("/root/reference/data")
np.savez_compressed(DATA / "materials_DEF.npz", **{f.name: h5io.read(f, "DEF") for f in sorted((REF / "materials").glob("*.h5"))})
with open(REF / "models/Musikverein_ConcertHall/model_export.json", "rb") as src, \
        gzip.GzipFile(DATA / "models" / "MV_model_export.json.gz", "wb", mtime=0) as dst:
    shutil.copyfileobj(src, dst)
print("wrote materials_DEF.npz and models/MV_model_export.json.gz")
