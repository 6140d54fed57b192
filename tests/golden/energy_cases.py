"""Scenes of the energy-diagnostic golden vectors (shared by make_golden_energy.py and the tests)."""
ENERGY_CASES = {
    "cart_lossy": dict(Nx=16, Ny=14, Nz=12, Nt=40, Nm=2, Mb=[2, 3], diff=False, sig="dhann30"),
    "cart_outside": dict(Nx=20, Ny=18, Nz=16, Nt=45, wall=6, Nm=1, Mb=2, src=[2, 2, 2], rcv=[[15, 13, 11], [2, 13, 2]],
                         diff=False, sig="hann10"),
    "fcc1_lossy": dict(Nx=16, Ny=14, Nz=12, Nt=40, fcc=True, Nm=2, Mb=[2, 3], diff=False, sig="dhann30"),
}


