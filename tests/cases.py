"""Shared parity cases: deterministic synthetic scenes (pffdtd_amd.synth) keyed by name."""
from pffdtd_amd import sim_data, synth

# name -> (shoebox kwargs, fcc_flag)
CASES = {
    "cart_lossy": (dict(Nx=24, Ny=22, Nz=20, Nt=60, Nm=2, Mb=[2, 3], rigid_every=7), 0),
    "cart_rigid": (dict(Nx=20, Ny=23, Nz=27, Nt=50, lossy=False), 0),
    "cart_mb11": (dict(Nx=26, Ny=20, Nz=22, Nt=80, Nm=3, Mb=[11, 1, 12], sig="dhann30"), 0),
    "cart_oddz": (dict(Nx=19, Ny=21, Nz=37, Nt=40, Nm=1, Mb=2), 0),
    # source OUTSIDE the box: the wave runs around it and into the ABC shell / ghost flips on every face
    "cart_outside": (dict(Nx=30, Ny=28, Nz=26, Nt=70, wall=7, Nm=2, Mb=[3, 5], src=[3, 3, 3],
                          rcv=[[24, 22, 20], [2, 23, 3], [12, 12, 12]]), 0),
    "cart_outside_oddz": (dict(Nx=23, Ny=25, Nz=35, Nt=60, wall=6, Nm=1, Mb=2, src=[2, 19, 29],
                               rcv=[[18, 2, 3], [2, 2, 2]]), 0),
    "cart_free": (dict(Nx=18, Ny=20, Nz=22, Nt=50, box=False, lossy=False), 0),
    "cart_wall2": (dict(Nx=20, Ny=18, Nz=22, Nt=40, wall=2, Nm=1, Mb=3), 0),  # boundary nodes in the ABC shell
    "fcc1_outside": (dict(Nx=30, Ny=28, Nz=26, Nt=60, fcc=True, wall=7, Nm=2, Mb=[2, 3], src=[3, 3, 3],
                          rcv=[[24, 22, 20], [2, 23, 3]]), 1),
    "fcc2_outside": (dict(Nx=30, Ny=32, Nz=26, Nt=60, fcc=True, wall=7, Nm=2, Mb=[2, 3], src=[3, 3, 3],
                          rcv=[[24, 26, 20], [2, 27, 3], [20, 3, 21]]), 2),
    "fcc1_lossy": (dict(Nx=24, Ny=22, Nz=20, Nt=60, fcc=True, Nm=2, Mb=[2, 3], rigid_every=5), 1),
    "fcc2_lossy": (dict(Nx=24, Ny=28, Nz=20, Nt=60, fcc=True, Nm=2, Mb=[2, 3], rigid_every=5), 2),
    "fcc2_mb11": (dict(Nx=22, Ny=24, Nz=26, Nt=70, fcc=True, Nm=2, Mb=[11, 4]), 2),
}


def make_sim(name, **override):
    kw, flag = CASES[name]
    kw = dict(kw)
    kw.update(override)
    sim = synth.shoebox(**kw)
    if flag == 2:
        synth.fold_fcc(sim)
        synth.sort_sim(sim)
    return sim


def make_sd(name, precision, scale=True, **override):
    sd = sim_data.SimData.from_sim(make_sim(name, **override), precision)
    if scale:
        sd.scale_input()
    return sd
