// pf_probe_kernels.h -- research / calibration kernels, NOT part of libpffdtd_hip.so (tools/libpf_probe.so only):
//   k_march_stream / k_linear_stream : the marching access pattern with the stencil taken out (tools/membench.py)
//   k_tb2_proto                      : LDS-based two-steps-per-pass prototype (tools/tb2_probe.py)
//   k_tb2_lds                        : k_tb2_reg with the y-halo rows exchanged through LDS (tools/tb2_probe.py)
// Tiling of k_tb2_proto: a workgroup owns TYE rows x 256 columns and marches x.  Stage 1 computes T = u^{n+1} of plane
// x+1 on rows 1..TYE-2 of the tile, stage 2 computes u^{n+2} of plane x on rows 2..TYE-3, columns 4..251.  Tiles overlap
// by 4 rows and 8 columns; u^n planes x..x+2 and u^{n+1} planes x-1..x+1 live in LDS rings.
#pragma once
#include "pf_air_fused.h"
#include "pf_tb2.h"

namespace pf {

// ---- calibration kernels (tools/membench.py): the marching access pattern with the stencil taken out -----------
// u0[cell] += u1[cell] over the interior, tiles and x-chunks exactly like k_air_cart_lean (R rows x 16 B per lane,
// WY waves in y); PF = how many planes ahead the loads are issued.  Tells the access pattern's own ceiling apart
// from what the stencil kernels lose on top of it.
template <typename Real, int R, int WY, int PF, int MODE = 0, int WZ = 1>
__global__ __launch_bounds__(64 * WY * WZ) void k_march_stream(const Real *__restrict__ u1, Real *__restrict__ u0, LeanParams fp) {
   typedef typename VecOf<Real>::type vec;
   constexpr int V = VecOf<Real>::V;
   const uint32_t total = (uint32_t)fp.nzt * fp.nyt * fp.nxc;
   uint32_t b = blockIdx.x;
   if (fp.swizzle) b = xcd_swizzle(b, total);
   const int zt = b % fp.nzt, yt = (b / fp.nzt) % fp.nyt, xc = b / (fp.nzt * fp.nyt);
   const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
   const int w = wv / WZ, wz = wv % WZ;
   const int z0 = ((zt * WZ + wz) * 64 + lane) * V;
   if (z0 >= fp.P) return;
   const int y0 = 1 + (yt * WY + w) * R;
   const int xs = fp.x_begin + xc * fp.chunk, xe = min(xs + fp.chunk, fp.x_end);
   uint32_t so[R];
   bool valid[R];
#pragma unroll
   for (int r = 0; r < R; r++) {
      so[r] = (uint32_t)min(y0 + r, fp.Ny - 1) * (uint32_t)fp.P + (uint32_t)z0;
      valid[r] = (y0 + r <= fp.Ny - 2);
   }
   vec a[PF + 1][R], o[PF + 1][R];
#pragma unroll
   for (int k = 0; k < PF; k++)
#pragma unroll
      for (int r = 0; r < R; r++) {
         const int x = min(xs + k, xe - 1);
         a[k][r] = (MODE & 1) ? __builtin_nontemporal_load((const vec *)(u1 + (int64_t)x * fp.plane + so[r])) : *(const vec *)(u1 + (int64_t)x * fp.plane + so[r]);
         o[k][r] = (MODE & 2) ? __builtin_nontemporal_load((const vec *)(u0 + (int64_t)x * fp.plane + so[r])) : *(const vec *)(u0 + (int64_t)x * fp.plane + so[r]);
      }
   for (int x = xs; x < xe; x++) {
      const int xl = min(x + PF, xe - 1);
#pragma unroll
      for (int r = 0; r < R; r++) {
         a[PF][r] = (MODE & 1) ? __builtin_nontemporal_load((const vec *)(u1 + (int64_t)xl * fp.plane + so[r])) : *(const vec *)(u1 + (int64_t)xl * fp.plane + so[r]);
         o[PF][r] = (MODE & 2) ? __builtin_nontemporal_load((const vec *)(u0 + (int64_t)xl * fp.plane + so[r])) : *(const vec *)(u0 + (int64_t)xl * fp.plane + so[r]);
      }
#pragma unroll
      for (int r = 0; r < R; r++)
         if (valid[r]) {
            const vec res = a[0][r] + o[0][r];
            if (MODE & 4) __builtin_nontemporal_store(res, (vec *)(u0 + (int64_t)x * fp.plane + so[r]));
            else *(vec *)(u0 + (int64_t)x * fp.plane + so[r]) = res;
         }
#pragma unroll
      for (int k = 0; k < PF; k++)
#pragma unroll
         for (int r = 0; r < R; r++) { a[k][r] = a[k + 1][r]; o[k][r] = o[k + 1][r]; }
   }
}
// plain linear stream over the same bytes: u0[i] += u1[i], 16 B per lane.
// MODE bit0: nontemporal loads, bit1: nontemporal stores, bit2: one-shot (no grid-stride loop), UNR vectors per thread
template <typename Real, int MODE, int UNR>
__global__ void k_linear_stream(const Real *__restrict__ u1, Real *__restrict__ u0, int64_t nvec) {
   typedef typename VecOf<Real>::type vec;
   const int64_t stride = (MODE & 4) ? 0 : (int64_t)gridDim.x * blockDim.x * UNR;
   // MODE bits 3..: log2(number of interleaved streams): block b works on stream b%S, position b/S
   const int S = 1 << (MODE >> 3);
   const int64_t blk = (S == 1) ? (int64_t)blockIdx.x : (int64_t)(blockIdx.x % S) * (gridDim.x / S) + blockIdx.x / S;
   for (int64_t i0 = blk * blockDim.x * UNR + threadIdx.x; i0 < nvec; i0 += stride) {
      vec a[UNR], o[UNR];
#pragma unroll
      for (int k = 0; k < UNR; k++) {
         const int64_t i = i0 + (int64_t)k * blockDim.x;
         if (i < nvec) {
            a[k] = (MODE & 1) ? __builtin_nontemporal_load((const vec *)u1 + i) : ((const vec *)u1)[i];
            o[k] = (MODE & 1) ? __builtin_nontemporal_load((const vec *)u0 + i) : ((const vec *)u0)[i];
         }
      }
#pragma unroll
      for (int k = 0; k < UNR; k++) {
         const int64_t i = i0 + (int64_t)k * blockDim.x;
         if (i < nvec) {
            const vec r = a[k] + o[k];
            if (MODE & 2) __builtin_nontemporal_store(r, (vec *)u0 + i); else ((vec *)u0)[i] = r;
         }
      }
      if (MODE & 4) break;
   }
}


template <int TYE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_tb2_proto(Tb2Params tp, float a1, float a2) {
   typedef f32x4 vec;
   constexpr int W = 256;
   constexpr int LROW = W + 8; // 4 pad floats each side so that column -1 / W reads stay in the row
   __shared__ __attribute__((aligned(16))) float Bs[3][TYE][LROW];
   __shared__ __attribute__((aligned(16))) float Ts[3][TYE][LROW];
   const uint32_t b = blockIdx.x;
   const int zt = b % tp.nzt, yt = (b / tp.nzt) % tp.nyt, xc = b / (tp.nzt * tp.nyt);
   const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
   const int ze0 = tp.z_begin - 4 + zt * (W - 8);      // first column of the extended tile
   const int ye0 = tp.y_begin - 2 + yt * (TYE - 4);    // first row of the extended tile
   const int xs = tp.x_begin + xc * tp.chunk, xe = min(xs + tp.chunk, tp.x_end);
   const int P = tp.P;
   const int64_t plane = tp.plane;
   const int zc = min(max(ze0 + lane * 4, 0), P - 4);  // clamped load column (tiles at the grid edge)
   auto grow = [&](int r) { return (int64_t)min(max(ye0 + r, 0), tp.Ny - 1) * P + zc; };

   auto fill_B = [&](int x, int slot) { // u^n plane x, all TYE rows of the tile
      const float *pl = (const float *)tp.B + (int64_t)x * plane;
      for (int r = w; r < TYE; r += WAVES) *(vec *)&Bs[slot][r][4 + lane * 4] = *(const vec *)(pl + grow(r));
   };
   auto stage1 = [&](int x, bool write_c) { // T(plane x) from B planes x-1, x, x+1 (slots (x-1)%3 ...) and A plane x; rows 1..TYE-2
      const float *pa = (const float *)tp.A + (int64_t)x * plane;
      float *pc = (float *)tp.C + (int64_t)x * plane;
      float(*Bm)[LROW] = Bs[(x + 2) % 3], (*Bc)[LROW] = Bs[x % 3], (*Bp)[LROW] = Bs[(x + 1) % 3];
      for (int r = 1 + w; r <= TYE - 2; r += WAVES) {
         const int col = 4 + lane * 4;
         const vec c = *(const vec *)&Bc[r][col];
         const vec yp = *(const vec *)&Bc[r + 1][col], ym = *(const vec *)&Bc[r - 1][col];
         const vec xp = *(const vec *)&Bp[r][col], xm = *(const vec *)&Bm[r][col];
         const float lf = Bc[r][col - 1], rt = Bc[r][col + 4];
         const vec old = *(const vec *)(pa + grow(r));
         vec o;
#pragma unroll
         for (int i = 0; i < 4; i++) {
            const float zp = (i == 3) ? rt : c[i < 3 ? i + 1 : 3];
            const float zm = (i == 0) ? lf : c[i > 0 ? i - 1 : 0];
            float p = a1 * c[i] - old[i];
            p = p + a2 * xp[i]; p = p + a2 * xm[i]; p = p + a2 * yp[i]; p = p + a2 * ym[i]; p = p + a2 * zp; p = p + a2 * zm;
            o[i] = p;
         }
         *(vec *)&Ts[x % 3][r][col] = o;
         // core cells of this tile own the C (u^{n+1}) output
         const bool core_row = (r >= 2 && r <= TYE - 3) && (ye0 + r >= tp.y_begin) && (ye0 + r < tp.Ny - tp.y_begin);
         const bool core_col = (lane >= 1 && lane <= 62) && (ze0 + lane * 4 + 3 < tp.Nz - tp.z_begin);
         if (write_c && core_row && core_col) __builtin_nontemporal_store(o, (vec *)(pc + grow(r)));
      }
   };
   auto stage2 = [&](int x) { // D(plane x) from T planes x-1, x, x+1 and B plane x; rows 2..TYE-3, lanes 1..62
      float *pd = (float *)tp.D + (int64_t)x * plane;
      float(*Tm)[LROW] = Ts[(x + 2) % 3], (*Tc)[LROW] = Ts[x % 3], (*Tp)[LROW] = Ts[(x + 1) % 3];
      float(*Bc)[LROW] = Bs[x % 3];
      for (int r = 2 + w; r <= TYE - 3; r += WAVES) {
         const int col = 4 + lane * 4;
         const vec c = *(const vec *)&Tc[r][col];
         const vec yp = *(const vec *)&Tc[r + 1][col], ym = *(const vec *)&Tc[r - 1][col];
         const vec xp = *(const vec *)&Tp[r][col], xm = *(const vec *)&Tm[r][col];
         const float lf = Tc[r][col - 1], rt = Tc[r][col + 4];
         const vec old = *(const vec *)&Bc[r][col];
         vec o;
#pragma unroll
         for (int i = 0; i < 4; i++) {
            const float zp = (i == 3) ? rt : c[i < 3 ? i + 1 : 3];
            const float zm = (i == 0) ? lf : c[i > 0 ? i - 1 : 0];
            float p = a1 * c[i] - old[i];
            p = p + a2 * xp[i]; p = p + a2 * xm[i]; p = p + a2 * yp[i]; p = p + a2 * ym[i]; p = p + a2 * zp; p = p + a2 * zm;
            o[i] = p;
         }
         const bool ok_row = (ye0 + r >= tp.y_begin) && (ye0 + r < tp.Ny - tp.y_begin);
         if (ok_row && lane >= 1 && lane <= 62 && ze0 + lane * 4 + 3 < tp.Nz - tp.z_begin) __builtin_nontemporal_store(o, (vec *)(pd + grow(r)));
      }
   };
   // D planes [xs, xe) need T planes xs-1 .. xe, which need B planes xs-2 .. xe+1
   fill_B(xs - 2, (xs - 2) % 3);
   fill_B(xs - 1, (xs - 1) % 3);
   fill_B(xs, xs % 3);
   __syncthreads();
   stage1(xs - 1, false);
   __syncthreads();
   for (int x = xs; x <= xe; x++) {
      fill_B(x + 1, (x + 1) % 3);    // overwrites the slot of plane x-2 (no longer needed)
      __syncthreads();
      stage1(x, x < xe);              // T(x) -> Ts[x%3]; this chunk owns C planes [xs, xe)
      __syncthreads();
      if (x - 1 >= xs) stage2(x - 1); // D(x-1) from T x-2, x-1, x and B x-1
      __syncthreads();
   }
}


// ---------------------------------------------------------------------------------------------------------------
// k_tb2_lds -- k_tb2_reg with the y-halo rows of u^n and u^{n-1} exchanged between the waves of a workgroup through
// LDS instead of being re-read through L1: a wave loads its own R rows of a plane from global memory (the top / bottom
// wave of the workgroup also the two rows beyond it), and one iteration later -- when the plane is first needed with
// halos -- publishes them to a double-buffered LDS tile, one barrier, and picks up its neighbours' rows.
// Global row loads per workgroup and plane: WY*R+4 (u^n) + WY*R+2 (u^{n-1}) instead of WY*(2R+6).
// ---------------------------------------------------------------------------------------------------------------
template <int R, int WY>
__global__ __launch_bounds__(64 * WY) void k_tb2_lds(Tb2Params tp, float a1, float a2) {
   typedef f32x4 vec;
   __shared__ __attribute__((aligned(16))) float sB[2][WY * R][256];
   __shared__ __attribute__((aligned(16))) float sA[2][WY * R][256];
   const uint32_t b = blockIdx.x;
   const int zt = b % tp.nzt, yt = (b / tp.nzt) % tp.nyt, xc = b / (tp.nzt * tp.nyt);
   const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
   const int ze0 = tp.z_begin - 4 + zt * 248;
   const int yo = tp.y_begin + (yt * WY + w) * R;
   const int xs = tp.x_begin + xc * tp.chunk, xe = min(xs + tp.chunk, tp.x_end);
   const int P = tp.P;
   const int64_t plane = tp.plane;
   const int zc = min(max(ze0 + lane * 4, 0), P - 4);
   int64_t offB[R + 4];
#pragma unroll
   for (int i = 0; i < R + 4; i++) offB[i] = (int64_t)min(max(yo - 2 + i, 0), tp.Ny - 1) * P + zc;
   const int z_end = tp.z_end ? tp.z_end : tp.Nz - tp.z_begin, y_end = tp.y_end ? tp.y_end : tp.Ny - tp.y_begin;
   const bool core_col = (lane >= 1 && lane <= 62) && (ze0 + lane * 4 + 3 < z_end);
   bool core_row[R];
#pragma unroll
   for (int r = 0; r < R; r++) core_row[r] = (yo + r < y_end);
   const bool top = (w == 0), bot = (w == WY - 1);

   // own rows (+ the rows beyond the workgroup for its first / last wave) of a plane from global memory
   auto loadB_own = [&](int x, vec *d) {
      const float *pl = (const float *)tp.B + (int64_t)x * plane;
#pragma unroll
      for (int r = 0; r < R; r++) d[r + 2] = *(const vec *)(pl + offB[r + 2]);
      if (top) { d[0] = *(const vec *)(pl + offB[0]); d[1] = *(const vec *)(pl + offB[1]); }
      if (bot) { d[R + 2] = *(const vec *)(pl + offB[R + 2]); d[R + 3] = *(const vec *)(pl + offB[R + 3]); }
   };
   auto loadA_own = [&](int x, vec *d) { // d: rows yo-1 .. yo+R
      const float *pl = (const float *)tp.A + (int64_t)x * plane;
#pragma unroll
      for (int r = 0; r < R; r++) d[r + 1] = *(const vec *)(pl + offB[r + 2]);
      if (top) d[0] = *(const vec *)(pl + offB[1]);
      if (bot) d[R + 1] = *(const vec *)(pl + offB[R + 2]);
   };
   // halo rows of the planes held in Bv (R+4 rows) and Av (R+2 rows) from the neighbouring waves
   auto exchange = [&](int slot, vec *Bv, vec *Av) {
#pragma unroll
      for (int r = 0; r < R; r++) {
         *(vec *)&sB[slot][w * R + r][lane * 4] = Bv[r + 2];
         *(vec *)&sA[slot][w * R + r][lane * 4] = Av[r + 1];
      }
      __syncthreads();
      if (!top) {
         Bv[0] = *(const vec *)&sB[slot][w * R - 2][lane * 4];
         Bv[1] = *(const vec *)&sB[slot][w * R - 1][lane * 4];
         Av[0] = *(const vec *)&sA[slot][w * R - 1][lane * 4];
      }
      if (!bot) {
         Bv[R + 2] = *(const vec *)&sB[slot][w * R + R][lane * 4];
         Bv[R + 3] = *(const vec *)&sB[slot][w * R + R + 1][lane * 4];
         Av[R + 1] = *(const vec *)&sA[slot][w * R + R][lane * 4];
      }
   };
   auto stencil = [&](const vec &c, const vec &xp, const vec &xm, const vec &yp, const vec &ym, const vec &old) {
      const float lf = lane_from_lower<true>(c[3]);
      const float rt = lane_from_upper<true>(c[0]);
      vec o;
#pragma unroll
      for (int i = 0; i < 4; i++) {
         const float zp = (i == 3) ? rt : c[i < 3 ? i + 1 : 3];
         const float zm = (i == 0) ? lf : c[i > 0 ? i - 1 : 0];
         float p = a1 * c[i] - old[i];
         p = p + a2 * xp[i]; p = p + a2 * xm[i]; p = p + a2 * yp[i]; p = p + a2 * ym[i]; p = p + a2 * zp; p = p + a2 * zm;
         o[i] = p;
      }
      return o;
   };

   vec Bp[R + 2], Bc[R + 4], Bn[R + 4], Bnn[R + 4], Ar[R + 2], Arn[R + 2];
   vec vm[R], vc[R + 2], vn[R + 2];
   {  // prologue: planes xs-2 (rows R+2 needed), xs-1, xs of u^n and xs-1 of u^{n-1}, halos through the same exchange
      vec t[R + 4], ta[R + 2];
      loadB_own(xs - 2, t);
      loadA_own(xs - 1, ta);
      exchange(0, t, ta);
#pragma unroll
      for (int j = 0; j < R + 2; j++) { Bp[j] = t[j + 1]; Ar[j] = ta[j]; }
      loadB_own(xs - 1, Bc);
      loadA_own(xs - 1, ta);
      exchange(1, Bc, ta);
      loadB_own(xs, Bn);       // its halos arrive at the top of the first iteration, together with Arn's
      loadA_own(xs, Arn);      // (u^{n-1} plane xs: used from the second iteration on)
   }
#pragma unroll
   for (int r = 0; r < R; r++) vm[r] = vec{0, 0, 0, 0};
#pragma unroll
   for (int j = 0; j < R + 2; j++) vc[j] = vec{0, 0, 0, 0};
   int it = 0;
   for (int x1 = xs - 1; x1 <= xe; x1++, it++) {
      // Bn = u^n plane x1+1 and Arn = u^{n-1} plane x1+1 arrived during the previous turn: complete them with halos
      exchange(it & 1, Bn, Arn);
      if (x1 < xe) loadB_own(x1 + 2, Bnn);
      // stage 1: u^{n+1}(x1) on rows yo-1 .. yo+R
#pragma unroll
      for (int j = 0; j < R + 2; j++) vn[j] = stencil(Bc[j + 1], Bn[j + 1], Bp[j], Bc[j + 2], Bc[j], Ar[j]);
      if (x1 >= xs && x1 < xe) {
         float *pc = (float *)tp.C + (int64_t)x1 * plane;
#pragma unroll
         for (int r = 0; r < R; r++)
            if (core_col && core_row[r]) __builtin_nontemporal_store(vn[r + 1], (vec *)(pc + offB[r + 2]));
      }
      if (x1 - 1 >= xs) {
         float *pd = (float *)tp.D + (int64_t)(x1 - 1) * plane;
#pragma unroll
         for (int r = 0; r < R; r++) {
            const vec o = stencil(vc[r + 1], vn[r + 1], vm[r], vc[r + 2], vc[r], Bp[r + 1]);
            if (core_col && core_row[r]) __builtin_nontemporal_store(o, (vec *)(pd + offB[r + 2]));
         }
      }
#pragma unroll
      for (int r = 0; r < R; r++) vm[r] = vc[r + 1];
#pragma unroll
      for (int j = 0; j < R + 2; j++) { vc[j] = vn[j]; Bp[j] = Bc[j + 1]; Ar[j] = Arn[j]; }
#pragma unroll
      for (int i = 0; i < R + 4; i++) { Bc[i] = Bn[i]; Bn[i] = Bnn[i]; }
      if (x1 + 1 < xe) loadA_own(x1 + 2, Arn);
   }
}


} // namespace pf
