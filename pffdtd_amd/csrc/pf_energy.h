// pf_energy.h -- device side of the energy-conservation diagnostic of the reference Python engine
// (python/fdtd/sim_fdtd.py:587-620,841-856): per-step sums that make up H_tot, E_lost and E_in.  Diagnostic
// path only (enabled with pf_opts.energy): it runs the unfused kernel sequence, keeps an explicit Laplacian grid
// like the reference does (Lu1, sim_fdtd.py:166,601-602) and reduces in double with one atomicAdd per wave.
#pragma once
#include "pf_kernels.h"

namespace pf {

// accumulator slots
enum { EN_INT = 0, EN_ABC = 1, EN_STORED = 2, EN_LOSS = 3, EN_ABCLOSS = 4, EN_IN = 5, EN_NACC = 8 };

__device__ __forceinline__ void wave_accumulate(double v, double *slot) {
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
   if ((threadIdx.x & 63) == 0) atomicAdd(slot, v);
}

// Laplacian of u1 on non-masked interior cells: nb_stencil_air_cart / _fcc (sim_fdtd.py:699-733)
template <typename Real, bool FCC>
__global__ void k_lap_air(const Real *__restrict__ u1, Real *__restrict__ Lu, const uint8_t *__restrict__ mask,
                          int64_t Nx, int64_t Ny, int64_t Nz, int64_t P, int64_t plane) {
   const int64_t iz = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   const int64_t iy = 1 + blockIdx.y, ix = 1 + blockIdx.z;
   if (iz < 1 || iz > Nz - 2 || iy > Ny - 2 || ix > Nx - 2) return;
   const int64_t jj = ix * plane + iy * P + iz;
   if ((mask[jj >> 3] >> (jj & 7)) & 1) return;
   if (!FCC) {
      Lu[jj] = Real(-6.0) * u1[jj] + u1[jj + plane] + u1[jj - plane] + u1[jj + P] + u1[jj - P] + u1[jj + 1] + u1[jj - 1];
   } else {
      Lu[jj] = Real(0.25) * (Real(-12.0) * u1[jj] + u1[jj + plane + P] + u1[jj - plane - P] + u1[jj + P + 1] + u1[jj - P - 1] +
                             u1[jj + plane + 1] + u1[jj - plane - 1] + u1[jj + plane - P] + u1[jj - plane + P] +
                             u1[jj + P - 1] + u1[jj - P + 1] + u1[jj + plane - 1] + u1[jj - plane + 1]);
   }
}
// adjacency-weighted Laplacian at boundary nodes: nb_stencil_bn_cart / _fcc (sim_fdtd.py:736-770)
template <typename Real, bool FCC>
__global__ void k_lap_bn(const Real *__restrict__ u1, Real *__restrict__ Lu, const int64_t *__restrict__ idx,
                         const uint16_t *__restrict__ adjv, int64_t P, int64_t plane, int64_t n) {
   const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (i >= n) return;
   const int64_t ii = idx[i];
   const uint32_t adj = adjv[i];
   Real s = -(Real)__popc(adj) * u1[ii];
   if (!FCC) {
      const int64_t off[6] = {plane, -plane, P, -P, 1, -1};
#pragma unroll
      for (int j = 0; j < 6; j++) s += (Real)((adj >> j) & 1u) * u1[ii + off[j]];
      Lu[ii] = s;
   } else {
      const int64_t off[12] = {plane + P, -plane - P, P + 1, -P - 1, plane + 1, -plane - 1,
                               plane - P, -plane + P, P - 1, -P + 1, plane - 1, -plane + 1};
#pragma unroll
      for (int j = 0; j < 12; j++) s += (Real)((adj >> j) & 1u) * u1[ii + off[j]];
      Lu[ii] = Real(0.25) * s;
   }
}
// S_int = sum over the interior of ((u1-u2)^2/l2 - u1*Lu2): nb_energy_int (sim_fdtd.py:841-844)
template <typename Real>
__global__ void k_energy_int(const Real *__restrict__ u1, const Real *__restrict__ u2, const Real *__restrict__ Lu2,
                             int64_t Nx, int64_t Ny, int64_t Nz, int64_t P, int64_t plane, double l2, double *acc) {
   const int64_t iz = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   const int64_t iy = 1 + blockIdx.y, ix = 1 + blockIdx.z;
   double e = 0.0;
   if (iz >= 1 && iz <= Nz - 2 && iy <= Ny - 2 && ix <= Nx - 2) {
      const int64_t jj = ix * plane + iy * P + iz;
      const double a = (double)u1[jj], b = (double)u2[jj], L = (double)Lu2[jj];
      e = ((a - b) * (a - b)) / l2 - a * L;
   }
   wave_accumulate(e, acc + EN_INT);
}
// S_abc = sum over ABC nodes of (1 - 2^-Q) * (same integrand) (sim_fdtd.py:595)
template <typename Real>
__global__ void k_energy_abc(const Real *__restrict__ u1, const Real *__restrict__ u2, const Real *__restrict__ Lu2,
                             const int64_t *__restrict__ idx, const int8_t *__restrict__ Q, int64_t n, double l2, double *acc) {
   const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   double e = 0.0;
   if (i < n) {
      const int64_t jj = idx[i];
      const double a = (double)u1[jj], b = (double)u2[jj], L = (double)Lu2[jj];
      e = (1.0 - exp2(-(double)Q[i])) * (((a - b) * (a - b)) / l2 - a * L);
   }
   wave_accumulate(e, acc + EN_ABC);
}
// S_stored = sum over lossy nodes of ssaf * sum_m (vh1^2*D + (Ts*gh1)^2*F): nb_energy_stored (sim_fdtd.py:847-849)
template <typename Real>
__global__ void k_energy_stored(const Real *__restrict__ vh1, const Real *__restrict__ gh1, const Real *__restrict__ ssaf,
                                const int8_t *__restrict__ mat, const int8_t *__restrict__ Mb, const double *__restrict__ DEF,
                                int64_t Nbl, double Ts, double *acc) {
   const int64_t nb = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   double e = 0.0;
   if (nb < Nbl) {
      const int k = mat[nb];
      double s = 0.0;
      for (int m = 0; m < Mb[k]; m++) {
         const double v = (double)vh1[st_idx(m, nb)], g = (double)gh1[st_idx(m, nb)];
         const double D = DEF[(k * 12 + m) * 3 + 0], F = DEF[(k * 12 + m) * 3 + 2];
         s += (v * v) * D + ((Ts * g) * (Ts * g)) * F;
      }
      e = (double)ssaf[nb] * s;
   }
   wave_accumulate(e, acc + EN_STORED);
}
// S_abcloss = sum over ABC nodes of (2^-Q * Q) * (u0 - u2ba)^2 after the update (sim_fdtd.py:617)
template <typename Real>
__global__ void k_energy_abcloss(const Real *__restrict__ u0, const Real *__restrict__ u2ba, const int64_t *__restrict__ idx,
                                 const int8_t *__restrict__ Q, int64_t n, double *acc) {
   const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   double e = 0.0;
   if (i < n) {
      const double d = (double)u0[idx[i]] - (double)u2ba[i];
      e = (exp2(-(double)Q[i]) * (double)Q[i]) * d * d;
   }
   wave_accumulate(e, acc + EN_ABCLOSS);
}
// source nodes: save u0 before the step, and S_in = sum (u0_after - u2in) * in_sig (sim_fdtd.py:590,620)
template <typename Real>
__global__ void k_energy_in(const Real *__restrict__ u0, Real *__restrict__ u2in, const int64_t *__restrict__ idx,
                            const Real *__restrict__ in_sigs, int64_t Ns, int64_t Nt, int64_t n, int after, double *acc) {
   const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   double e = 0.0;
   if (s < Ns) {
      if (!after) u2in[s] = u0[idx[s]];
      else e = ((double)u0[idx[s]] - (double)u2in[s]) * (double)in_sigs[s * Nt + n];
   }
   if (after) wave_accumulate(e, acc + EN_IN);
}
// S_loss needs the old and the new branch currents: evaluated next to the FD update from a copy of vh1 taken before
template <typename Real>
__global__ void k_energy_loss(const Real *__restrict__ vh_old, const Real *__restrict__ vh_new, const Real *__restrict__ ssaf,
                              const int8_t *__restrict__ mat, const int8_t *__restrict__ Mb, const double *__restrict__ DEF,
                              int64_t Nbl, double *acc) {
   const int64_t nb = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   double e = 0.0;
   if (nb < Nbl) {
      const int k = mat[nb];
      double s = 0.0;
      for (int m = 0; m < Mb[k]; m++) {
         const double v = (double)vh_old[st_idx(m, nb)] + (double)vh_new[st_idx(m, nb)];
         s += (v * v) * DEF[(k * 12 + m) * 3 + 1];
      }
      e = (double)ssaf[nb] * s;
   }
   wave_accumulate(e, acc + EN_LOSS);
}

} // namespace pf
