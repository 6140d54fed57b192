#!/usr/bin/env python3
"""Writes pffdtd_amd/data/models/open_scene.json: a small hand-made export in the reference's JSON schema that exercises
what the CTK / Musikverein exports do not: unmarked `_RIGID` triangles (sidedness 0 -> material -1), an open scene
(no ceiling; custom bmin/bmax), a free-standing two-sided panel, a tilted one-sided reflector."""
import json
from pathlib import Path

HERE = Path(__file__).resolve().parent
DATA = HERE.parent.parent / "pffdtd_amd" / "data"


def quad(p0, p1, p2, p3):
    return [p0, p1, p2, p3], [[0, 1, 2], [0, 2, 3]]


def mat(quads, sides, color):
    pts, tris = [], []
    for q in quads:
        qp, qt = q
        o = len(pts)
        pts += qp
        tris += [[a + o, b + o, c + o] for a, b, c in qt]
    return {"pts": pts, "tris": tris, "sides": [sides] * len(tris), "color": color}


Lx, Ly, Lz = 4.1, 3.3, 2.6
floor = quad([0, 0, 0], [Lx, 0, 0], [Lx, Ly, 0], [0, Ly, 0])
walls = [quad([0, 0, 0], [0, Ly, 0], [0, Ly, Lz], [0, 0, Lz]), quad([Lx, 0, 0], [Lx, 0, Lz], [Lx, Ly, Lz], [Lx, Ly, 0]),
         quad([0, 0, 0], [0, 0, Lz], [Lx, 0, Lz], [Lx, 0, 0]), quad([0, Ly, 0], [Lx, Ly, 0], [Lx, Ly, Lz], [0, Ly, Lz])]
panel = [quad([1.3, 0.9, 0.4], [1.3, 2.1, 0.4], [1.3, 2.1, 1.9], [1.3, 0.9, 1.9])]
reflector = [quad([2.2, 0.5, 1.2], [3.6, 0.5, 2.0], [3.6, 2.6, 2.0], [2.2, 2.6, 1.2])]
scene = {"mats_hash": {"_RIGID": mat([floor], 0, [255, 255, 255]), "Brick": mat(walls, 2, [180, 80, 60]),
                       "Panel": mat(panel, 3, [60, 120, 200]), "Reflector": mat(reflector, 1, [90, 200, 90])},
         "sources": [{"xyz": [0.7, 0.8, 1.1], "name": "S1"}, {"xyz": [3.2, 2.4, 0.9], "name": "S2"}],
         "receivers": [{"xyz": [2.9, 1.1, 0.8], "name": "R1"}, {"xyz": [0.6, 2.6, 1.5], "name": "R2"}],
         "export_datetime": "hand-made"}
(DATA / "models" / "open_scene.json").write_text(json.dumps(scene))
print("wrote models/open_scene.json")
