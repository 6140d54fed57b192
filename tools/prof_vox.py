import sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
from pffdtd_amd import scenes, setup_io
from pffdtd_amd.room_geo import RoomGeo
from pffdtd_amd.voxelizer import VoxScene
cfg = scenes.CONFIGS["mv_fcc_gpu"]
rg = RoomGeo(str(scenes.model_path("MV")))
sc = setup_io.SimConsts(Tc=20, rh=50, fmax=cfg["fmax"], PPW=cfg["PPW"], fcc=True)
cg = setup_io.CartGrid(h=sc.h, offset=3.5, bmin=rg.bmin, bmax=rg.bmax, fcc=True)
vs = VoxScene(rg, cg, fcc=True)
pr = cProfile.Profile(); pr.enable(); vs.calc_adj(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
