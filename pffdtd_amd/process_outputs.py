"""Receiver post-processing: the reference's `ProcessOutputs` (python/fdtd/process_outputs.py:30-290)
on top of pffdtd_amd.h5io -- recombination of the 8-node receivers with `out_alpha`, integrator + low-cut (undoes the
source differentiation), optional symmetric low-pass, resampling, WAV / h5 export.

    python -m pffdtd_amd.process_outputs --data_dir D [--fcut_lowcut 10 --N_order_lowcut 4 --fcut_lowpass F
                                         --N_order_lowpass 8 --symmetric_lowpass --resample_Fs 48000 --save_wav
                                         --air_abs_filter stokes|modal|OLA]

Resampling: the reference calls resampy's `resample(..., filter='kaiser_best')`; resampy is not in this image, its
published algorithm and filter design are restated in pffdtd_amd/resample.py (parity unpinned, see there).  The
air-absorption filters are in pffdtd_amd/air_abs.py.
"""
import argparse
from pathlib import Path

import numpy as np
from scipy.signal import bilinear_zpk, butter, lfilter, sosfilt, zpk2sos

from . import h5io
from .resample import resample as resample_kaiser_best


class ProcessOutputs:
    def __init__(self, data_dir):
        data_dir = Path(data_dir)
        c = data_dir / "comms_out.h5"
        self.out_alpha = h5io.read(c, "out_alpha")
        self.Nr, self.Nt, self.diff = int(h5io.read(c, "Nr")), int(h5io.read(c, "Nt")), bool(h5io.read(c, "diff"))
        k = data_dir / "sim_consts.h5"
        self.Ts = float(h5io.read(k, "Ts"))
        self.Tc = float(h5io.read(k, "Tc")) if h5io.exists(k, "Tc") else None
        self.rh = float(h5io.read(k, "rh")) if h5io.exists(k, "rh") else None
        self.u_out = h5io.read(data_dir / "sim_outs.h5", "u_out")
        assert self.out_alpha.size == self.Nr and self.u_out.size == self.Nr * self.Nt and self.out_alpha.ndim == 2
        self.Fs = 1 / self.Ts
        self.Ts_f, self.Fs_f, self.Nt_f = self.Ts, self.Fs, self.Nt
        self.r_out = self.r_out_f = None
        self.data_dir = data_dir

    def print(self, fstring):
        print(f"--PROCESS_OUTPUTS: {fstring}")

    def initial_process(self, fcut=10.0, N_order=4):
        """Recombine (process_outputs.py:95), store r_out next to u_out (:97-104), integrate + low-cut (:106-127)."""
        oa = self.out_alpha
        self.r_out = np.sum((self.u_out * oa.reshape(-1)[:, None]).reshape((*oa.shape, -1)), axis=1)
        h5io.write(self.data_dir / "sim_outs.h5", "r_out", self.r_out, append=True)
        Ts = self.Ts
        if fcut > 0:
            if self.diff:
                z, p, k = butter(N_order, fcut * 2 * np.pi, btype="high", analog=True, output="zpk")
                assert np.all(z == 0.0)
                zd, pd, kd = bilinear_zpk(z[1:], p, k, 1 / Ts)  # one zero removed = integrator
                sos = zpk2sos(zd, pd, kd)
            else:
                sos = butter(N_order, 2 * Ts * fcut, btype="high", output="sos")
            self.r_out_f = sosfilt(sos, self.r_out)
        elif self.diff:
            self.r_out_f = lfilter(Ts / 2 * np.array([1, 1]), np.array([1, -1]) * 0 + np.array([1, 1]), self.r_out)
        else:
            self.r_out_f = np.copy(self.r_out)

    def apply_lowpass(self, fcut, N_order=8, symmetric=True):
        if symmetric:
            assert N_order % 2 == 0
            N_order //= 2
        sos = butter(N_order, 2 * self.Ts_f * fcut, btype="low", output="sos")
        r = sosfilt(sos, self.r_out_f)
        if symmetric:
            r = sosfilt(sos, r[:, ::-1])[:, ::-1]
        self.r_out_f = r

    def resample(self, Fs_f=48e3):
        if self.Fs == Fs_f:
            return
        self.print("resampling")
        self.r_out_f = resample_kaiser_best(self.r_out_f, self.Fs, Fs_f, axis=-1)  # process_outputs.py:161
        self.Fs_f = Fs_f
        self.Ts_f, self.Nt_f = 1 / Fs_f, self.r_out_f.shape[-1]

    def _air(self, name, fn, **kw):
        from . import air_abs
        if self.Tc is None or self.rh is None:
            raise ValueError("sim_consts.h5 must hold Tc and rh for the air-absorption filters")
        self.print(f"applying {name} air absorption filter")
        self.r_out_f = getattr(air_abs, fn)(self.r_out_f, self.Fs_f, Tc=self.Tc, rh=self.rh, **kw)
        self.Nt_f = self.r_out_f.shape[-1]  # the filters lengthen the responses

    def apply_stokes_filter(self, NdB=120):  # process_outputs.py:169-180
        self._air("Stokes'", "apply_visco_filter", NdB=NdB)

    def apply_modal_filter(self):  # :182-193
        self._air("modal", "apply_modal_filter")

    def apply_ola_filter(self):  # :195-206
        self._air("OLA", "apply_ola_filter")

    def save_wav(self):
        from scipy.io.wavfile import write as wavwrite
        r = np.atleast_2d(self.r_out_f)
        n_fac = np.max(np.abs(r))
        for i in range(r.shape[0]):
            wavwrite(self.data_dir / f"R{i + 1:03d}_out_normalised.wav", int(self.Fs_f), r[i] / n_fac)
            if n_fac < 1.0:
                wavwrite(self.data_dir / f"R{i + 1:03d}_out_native.wav", int(self.Fs_f), r[i])

    def save_h5(self):
        f = self.data_dir / "sim_outs_processed.h5"
        h5io.write(f, "r_out_f", self.r_out_f, append=False)
        h5io.write(f, "Fs_f", np.float64(self.Fs_f))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--data_dir", type=str, required=True)
    p.add_argument("--resample_Fs", type=float, default=48e3)
    p.add_argument("--fcut_lowcut", type=float, default=10.0)
    p.add_argument("--fcut_lowpass", type=float, default=0.0)
    p.add_argument("--N_order_lowcut", type=int, default=8)  # process_outputs.py:320
    p.add_argument("--N_order_lowpass", type=int, default=8)
    p.add_argument("--symmetric_lowpass", action="store_true")
    p.add_argument("--save_wav", action="store_true")
    p.add_argument("--plot", action="store_true", help="ignored (plotting is out of scope)")
    p.add_argument("--plot_raw", action="store_true", help="ignored (plotting is out of scope)")
    p.add_argument("--air_abs_filter", type=str, default="none", help="stokes, modal, OLA, or none")
    a = p.parse_args()
    po = ProcessOutputs(a.data_dir)
    po.initial_process(fcut=a.fcut_lowcut, N_order=a.N_order_lowcut)
    if a.resample_Fs:  # the reference resamples first and low-passes at the new rate (process_outputs.py:330-334)
        po.resample(a.resample_Fs)
    if a.fcut_lowpass > 0:
        po.apply_lowpass(fcut=a.fcut_lowpass, N_order=a.N_order_lowpass, symmetric=a.symmetric_lowpass)
    flt = a.air_abs_filter.lower()  # process_outputs.py:338-343
    if flt == "modal":
        po.apply_modal_filter()
    elif flt == "stokes":
        po.apply_stokes_filter()
    elif flt == "ola":
        po.apply_ola_filter()
    elif flt != "none":
        raise SystemExit("--air_abs_filter: stokes, modal, OLA or none")
    po.save_h5()
    if a.save_wav:
        po.save_wav()


if __name__ == "__main__":
    main()
