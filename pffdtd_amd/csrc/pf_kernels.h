// pf_kernels.h -- device code of the MI355X FDTD engine (gfx950 only, wave64).
//
// HBM layout ("padded file layout"): a state grid is Real[Nx][Ny][P], P = z pitch = Nz rounded up to a
// multiple of 128 B, so every row starts on a cache line and a lane's 16-byte vector never straddles rows.
// Linear index jj = (ix*Ny + iy)*P + iz.  The engine's own skip-mask has one bit per padded cell (bit jj&7 of
// byte jj>>3) and is the union of: boundary nodes (the reference's bn_mask, fdtd_data.h:567-572), the z ghost
// columns iz=0 / iz=Nz-1, the pad columns iz>=Nz, and -- for the FCC checkerboard form (fcc_flag 1) -- the
// odd-parity cells that do not exist on the subgrid (cpu_engine.h:200,219).
//
// Numerics: every kernel follows the operation order of the reference C CPU engine (file:line cited at each
// kernel); with SG=false nothing is contracted (the TU is built with -ffp-contract=off), so results are
// bit-identical to cpu_engine.h.  SG=true is the reference GPU engine's "safeguarded" arithmetic (see upd7 below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace pf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename Real> struct VecOf;
template <> struct VecOf<float> { typedef f32x4 type; static constexpr int V = 4; };
template <> struct VecOf<double> { typedef f64x2 type; static constexpr int V = 2; };

// ---- wave64 neighbour exchange along the unit-stride axis -------------------------------------------------
// DPP wave shifts (gfx9 family): wave_shr:1 moves data to the next-higher lane (lane i reads lane i-1),
// wave_shl:1 the other way.  Lanes with no source keep `old` (=0).
// (bound_ctrl: a lane without a source gets 0 from the instruction itself, no register to preset)
__device__ __forceinline__ int dpp_from_lower(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int dpp_from_upper(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }

template <bool DPP> __device__ __forceinline__ float lane_from_lower(float v) {
   if (DPP) return __int_as_float(dpp_from_lower(__float_as_int(v)));
   return __shfl_up(v, 1, 64);
}
template <bool DPP> __device__ __forceinline__ float lane_from_upper(float v) {
   if (DPP) return __int_as_float(dpp_from_upper(__float_as_int(v)));
   return __shfl_down(v, 1, 64);
}
template <bool DPP> __device__ __forceinline__ double lane_from_lower(double v) {
   if (DPP) {
      int lo = dpp_from_lower(__double2loint(v)), hi = dpp_from_lower(__double2hiint(v));
      return __hiloint2double(hi, lo);
   }
   return __shfl_up(v, 1, 64);
}
template <bool DPP> __device__ __forceinline__ double lane_from_upper(double v) {
   if (DPP) {
      int lo = dpp_from_upper(__double2loint(v)), hi = dpp_from_upper(__double2hiint(v));
      return __hiloint2double(hi, lo);
   }
   return __shfl_down(v, 1, 64);
}

// self-test kernel for the two DPP controls (run once per process; the engine falls back to ds_bpermute
// shuffles if the semantics are not the expected ones)
static __global__ void k_dpp_selftest(int *out) {
   int lane = threadIdx.x & 63;
   int a = dpp_from_lower(lane + 100);
   int b = dpp_from_upper(lane + 100);
   int ok = 1;
   if (lane > 0 && a != lane - 1 + 100) ok = 0;
   if (lane < 63 && b != lane + 1 + 100) ok = 0;
   unsigned long long all = __ballot(ok);
   if (lane == 0) out[0] = (all == ~0ull) ? 1 : 0;
}

// =============================================================================================================
// Numerics of the air and rigid-node updates (pf_opts.numerics), template flag SG:
//   SG = false, PF_NUM_CPU_EXACT: the reference C CPU engine's expression, accumulated left to right with separate
//     multiplies and adds (cpu_engine.h:175-223,234-287) -- bit-identical to it.
//   SG = true, PF_NUM_GPU_SAFEGUARDED: the reference GPU engine's arithmetic (fdtd_common.h:44-71, gpu_engine.h:220-274,
//     288-348): the neighbours are summed pairwise ("divide-conquer add"), in fp32 with every add rounded TOWARDS ZERO
//     (ADD_O = __fadd_rz: a sum that never rounds away from zero cannot pump energy into the scheme -- the reference's
//     long-run fp32 stability measure), then two round-to-nearest FMAs  c1*u1 + (c2*sum - u0).  In fp64 the same tree and
//     FMAs with round-to-nearest throughout (ADD_O = __dadd_rn).
// The fp32 tree runs inside ONE asm statement bracketed by two s_setreg writes of MODE.FP_ROUND[1:0] (3 = towards zero for
// single precision, 0 = nearest even): nothing else can be scheduled into the window, and the mode is back to the default
// before any other instruction issues.
// =============================================================================================================
__device__ __forceinline__ float fma_rn(float a, float b, float c) { return __builtin_fmaf(a, b, c); }   // FMA_D = __fmaf_rn
__device__ __forceinline__ double fma_rn(double a, double b, double c) { return __builtin_fma(a, b, c); } // FMA_D = __fma_rn
__device__ __forceinline__ float sg_sum6(float a0, float a1, float a2, float a3, float a4, float a5) {
   float t1, t2; // ((a0 + a1) + (a2 + a3)) + (a4 + a5), gpu_engine.h:230-234
   asm("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
       "v_add_f32 %0, %2, %3\n\t"
       "v_add_f32 %1, %4, %5\n\t"
       "v_add_f32 %0, %0, %1\n\t"
       "v_add_f32 %1, %6, %7\n\t"
       "v_add_f32 %0, %0, %1\n\t"
       "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
       : "=&v"(t1), "=&v"(t2)
       : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5));
   return t1;
}
__device__ __forceinline__ double sg_sum6(double a0, double a1, double a2, double a3, double a4, double a5) {
   double t1 = a0 + a1, t2 = a2 + a3;
   t1 = t1 + t2;
   t2 = a4 + a5;
   return t1 + t2;
}
// 13-point: n[0..11] in the adjacency-bit order (+x+y)(-x-y)(+y+z)(-y-z)(+x+z)(-x-z)(+x-y)(-x+y)(+y-z)(-y+z)(+x-z)(-x+z);
// tree of gpu_engine.h:257-267
__device__ __forceinline__ float sg_sum12(const float (&n)[12]) {
   float t1, t2, t3, t4;
   asm("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
       "v_add_f32 %0, %4, %5\n\t"    // t1 = n0 + n1
       "v_add_f32 %1, %6, %7\n\t"    // t2 = n2 + n3
       "v_add_f32 %0, %0, %1\n\t"    // t1 += t2
       "v_add_f32 %2, %8, %9\n\t"    // t3 = n4 + n5
       "v_add_f32 %3, %10, %11\n\t"  // t4 = n6 + n7
       "v_add_f32 %2, %2, %3\n\t"    // t3 += t4
       "v_add_f32 %1, %12, %13\n\t"  // t2 = n8 + n9
       "v_add_f32 %0, %0, %1\n\t"    // t1 += t2
       "v_add_f32 %3, %14, %15\n\t"  // t4 = n10 + n11
       "v_add_f32 %2, %2, %3\n\t"    // t3 += t4
       "v_add_f32 %0, %0, %2\n\t"    // t1 += t3
       "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
       : "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4)
       : "v"(n[0]), "v"(n[1]), "v"(n[2]), "v"(n[3]), "v"(n[4]), "v"(n[5]), "v"(n[6]), "v"(n[7]), "v"(n[8]), "v"(n[9]),
         "v"(n[10]), "v"(n[11]));
   return t1;
}
__device__ __forceinline__ double sg_sum12(const double (&n)[12]) {
   double t1 = n[0] + n[1], t2 = n[2] + n[3];
   t1 = t1 + t2;
   double t3 = n[4] + n[5], t4 = n[6] + n[7];
   t3 = t3 + t4;
   t2 = n[8] + n[9];
   t1 = t1 + t2;
   t4 = n[10] + n[11];
   t3 = t3 + t4;
   return t1 + t3;
}
// one 7-point air update; neighbours in the order +x, -x, +y, -y, +z, -z (file axes: x = slowest, z = unit stride)
template <bool SG, typename Real>
__device__ __forceinline__ Real upd7(Real a1, Real a2, Real c, Real old, Real xp, Real xm, Real yp, Real ym, Real zp, Real zm) {
   if (SG) return fma_rn(a1, c, fma_rn(a2, sg_sum6(xp, xm, yp, ym, zp, zm), -old)); // gpu_engine.h:235
   Real p = a1 * c - old; // cpu_engine.h:181-191
   p = p + a2 * xp; p = p + a2 * xm; p = p + a2 * yp; p = p + a2 * ym; p = p + a2 * zp; p = p + a2 * zm;
   return p;
}
// one 13-point air update, n[] in the adjacency-bit order
template <bool SG, typename Real> __device__ __forceinline__ Real upd13(Real a1, Real a2, Real c, Real old, const Real (&n)[12]) {
   if (SG) return fma_rn(a1, c, fma_rn(a2, sg_sum12(n), -old)); // gpu_engine.h:268
   Real p = a1 * c - old; // cpu_engine.h:200-218
#pragma unroll
   for (int k = 0; k < 12; k++) p = p + a2 * n[k];
   return p;
}
// rigid boundary node: centre coefficient 2 - sl2*K, neighbour k present when adjacency bit k is set
// (cpu_engine.h:234-287; gpu_engine.h:288-348: products bit*u1 exact, tree of towards-zero adds, two FMAs)
template <bool SG, int NN, typename Real>
__device__ __forceinline__ Real upd_rigid(Real a2, Real sl2, uint32_t adj, Real c, Real old, const Real (&nb)[NN]) {
   const Real two = 2.0, K = (Real)__popc(adj);
   if (SG) {
      Real w[NN];
#pragma unroll
      for (int k = 0; k < NN; k++) w[k] = (Real)((adj >> k) & 1u) * nb[k];
      const Real b1 = two - sl2 * K; // (gpu_engine.h:300: `_2 - csl2*K`)
      Real sum;
      if constexpr (NN == 6) sum = sg_sum6(w[0], w[1], w[2], w[3], w[4], w[5]); else sum = sg_sum12(w);
      return fma_rn(b1, c, fma_rn(a2, sum, -old));
   }
   const Real b1 = two - sl2 * K;
   Real p = b1 * c - old;
#pragma unroll
   for (int k = 0; k < NN; k++) {
      // a2 * (Real)bit (cpu_engine.h:247-252) as a select: a2 * 1 = a2 and a2 * 0 = +0 exactly (a2 > 0) -- one instruction instead of a
      // conversion and a product, none at all where the adjacency word is wave-uniform (the alike blocks of pf_wall.h: a scalar select)
      const Real wk = ((adj >> k) & 1u) ? a2 : Real(0);
      p = p + wk * nb[k];
   }
   return p;
}
// ABC loss u0 = (u0 + lQ*u2) / (1 + lQ).  CPU engine: the literal 1.0 makes denominator and division double even in the
// float build (cpu_engine.h:228) -- reproduced; GPU engine: all in Real (gpu_engine.h:351-365, `Real _1`)
template <bool SG, typename Real> __device__ __forceinline__ Real abc_loss(Real u, Real u2, Real lQ) {
   if (SG) { const Real one = 1.0; return (u + lQ * u2) / (one + lQ); }
   const Real num = u + lQ * u2;
   return (Real)((double)num / (1.0 + (double)lQ));
}

// ---- XCD-aware workgroup order (guide T1): hardware places block b on XCD b%8; give each XCD one contiguous
// run of logical tiles so that tiles sharing halo rows share an L2.  Bijective for any total.
__device__ __forceinline__ uint32_t xcd_swizzle(uint32_t b, uint32_t total) {
   uint32_t q = total >> 3, r = total & 7u;
   uint32_t k = b & 7u, i = b >> 3;
   return k * q + (k < r ? k : r) + i;
}

// Swizzle mode 2, "banded" (the default since round 2): the launch walks the x chunks in order, and inside a chunk XCD k
// (= blocks k, k+8, ...) works through the k-th contiguous band of that chunk's y-z tiles -- all XCDs stay on the same
// planes (their halo planes meet in the memory-side cache) and tiles that share halo rows share an L2.  The grid holds
// nxc * 8 * ceil(T/8) blocks (T = tiles per chunk), the padding blocks return at once.  Measured against mode 1 (one
// contiguous run of the whole launch per XCD, so the XCDs work on different x ranges): k_air_fcc 2.46 -> 2.31 ms at 1024^3,
// Musikverein 3.65 -> 3.31 ms per step of interior update.
__device__ __forceinline__ bool xcd_band(uint32_t blk, uint32_t T, uint32_t nxc, uint32_t &b) {
   const uint32_t per = (T + 7u) >> 3, span = per << 3;
   const uint32_t c = blk / span, r = blk % span, kx = r & 7u, p2 = r >> 3;
   const uint32_t j0 = (kx * T) >> 3, j1 = ((kx + 1u) * T) >> 3; // balanced bands: sizes differ by at most one tile
   if (p2 >= j1 - j0 || c >= nxc) return false;
   b = c * T + j0 + p2;
   return true;
}
__host__ __device__ inline uint32_t xcd_band_blocks(uint32_t T, uint32_t nxc) { return nxc * (((T + 7u) >> 3) << 3); }

struct AirParams {
   int64_t Ny, P, plane;  // rows, z pitch, plane stride Ny*P (elements)
   int32_t x_begin, x_end; // planes updated by this launch: [x_begin, x_end)
   int32_t chunk;          // planes marched per workgroup
   int32_t nzt, nyt, nxc;  // tile counts along z, y and x-chunks
   int32_t swizzle;
   // VG (virtual ghost shell + in-kernel ABC) only:
   int32_t Nx, Nz, first, last, fold;
   // 1: the grid is stored with the file's x and z axes exchanged (Engine::swz: unit stride along file x).  The kernels work in
   // storage coordinates; only the ORDER in which the neighbours enter the sums follows the file's axes, so that the bits are
   // the reference's whatever the storage order
   int32_t swz;
};

// =============================================================================================================
// Air update, 7-point Cartesian (cpu_engine.h:175-194; reference GPU counterpart gpu_engine.h:220-242).
//   u0 = a1*u1 - u0 + a2*(+x) + a2*(-x) + a2*(+y) + a2*(-y) + a2*(+z) + a2*(-z), accumulated left to right,
//   skipped where the mask bit is set.
// 2.5D register marching: a workgroup owns a (WY*R rows) x (WZ*64*V columns) tile and marches along x; each
// lane owns R rows x V consecutive z of the tile.  Per plane it issues one coalesced 16 B/lane load per row for
// the next u1 plane (+2 halo rows per wave), one for u0, one store.  x neighbours live in registers (prev /
// next), y neighbours are adjacent registers of the same lane (halo rows loaded from L2), z neighbours inside
// a lane's vector are registers and across lanes one DPP wave shift; the two wave-edge columns are loaded by
// the edge lanes.
// =============================================================================================================
// VG = true: the ghost shell is virtual (loads that would touch plane 0 / row 0 / column Nz-1 read their mirror
// cells, cf. pf_air_fused.h) and the ABC loss is applied in-kernel, so no flip / ABC kernels run around it.
// ABCK = true (without VG): only the ABC loss moves in-kernel; the ghost shell is still maintained in memory by the
// flip kernels (cheaper than VG's per-row patches for the 13-point kernel).
template <typename Real, int R, int WY, int WZ, bool SG, bool DPP, bool VG = false, bool ABCK = false, int LW = 64>
__global__ __launch_bounds__(64 * WY * WZ) void k_air_cart(const Real *__restrict__ u1, Real *__restrict__ u0,
                                                          const uint8_t *__restrict__ mask, Real a1, Real a2,
                                                          AirParams ap, Real labc,
                                                          Real *u0_dst = nullptr) { // u0_dst: write there instead of in place
   typedef typename VecOf<Real>::type vec;
   constexpr int V = VecOf<Real>::V;
   const uint32_t total = (uint32_t)ap.nzt * ap.nyt * ap.nxc;
   uint32_t b = blockIdx.x;
   if (ap.swizzle == 2) { if (!xcd_band(blockIdx.x, (uint32_t)ap.nzt * ap.nyt, (uint32_t)ap.nxc, b)) return; }
   else if (ap.swizzle) b = xcd_swizzle(b, total);
   const int zt = b % ap.nzt;
   const int yt = (b / ap.nzt) % ap.nyt;
   const int xc = b / (ap.nzt * ap.nyt);
   // LW lanes span a row segment; a wave stacks 64/LW such segments in y (LW < 64: narrow grids, cf. Engine::pick_lw)
   static_assert(LW == 64 || LW == 32 || LW == 16, "row segments are 64, 32 or 16 lanes wide");
   constexpr int NSUB = 64 / LW;
   const int wlane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const int lane = wlane % LW, sub = wlane / LW; // `lane` = position inside the row segment
   const int wz = wave % WZ, wy = wave / WZ;
   const int64_t z0 = ((int64_t)(zt * WZ + wz) * LW + lane) * V;
   const int64_t y0 = 1 + ((int64_t)(yt * WY + wy) * NSUB + sub) * R;
   const int xs = ap.x_begin + xc * ap.chunk;
   const int xe = min(xs + ap.chunk, ap.x_end);
   const int64_t Ny = ap.Ny, P = ap.P, plane = ap.plane;
   const bool active = z0 < P;
   if (y0 - (int64_t)sub * R > Ny - 2) return; // whole wave out of rows (uniform per wave)
   const int64_t zl = active ? z0 : 0; // inactive lanes read column 0 (in range), never store
   const bool need_l = (lane == 0) && (z0 > 0);
   const bool need_r = (lane == LW - 1) && (z0 + V < P);

   // row offsets inside a plane, clamped so that halo/invalid rows stay in range
   auto rowsrc = [&](int64_t y) -> int64_t {
      if (y > Ny - 1) y = Ny - 1;
      if (VG) { if (y == 0) return 2; if (y == Ny - 1) return Ny - 3; }
      return y;
   };
   auto planesrc = [&](int x) -> int64_t {
      if (VG) { if (ap.first && x == 0) return 2; if (ap.last && x == ap.Nx - 1) return ap.Nx - 3; }
      return x;
   };
   int64_t roff[R + 2], soff[R];
#pragma unroll
   for (int j = 0; j < R + 2; j++) roff[j] = rowsrc(y0 - 1 + j) * P + zl;
#pragma unroll
   for (int r = 0; r < R; r++) soff[r] = (y0 + r > Ny - 1 ? Ny - 1 : y0 + r) * P + zl; // true rows: old state, mask, store
   // virtual z ghost columns: per-lane constants (column 0 mirrors column 2, column Nz-1 mirrors Nz-3)
   const int zzN = VG ? (int)(ap.Nz - 1 - z0) : -1;
   const bool fix0 = VG && (z0 == 0);
   const bool fixR = VG && (zzN == V);
   uint32_t qzbits = 0;
   if (VG || ABCK) {
#pragma unroll
      for (int i = 0; i < V; i++)
         if (active && (z0 + i == 1 || z0 + i == ap.Nz - 2)) qzbits |= 1u << i;
   }
   const bool wave_has_qz = (VG || ABCK) && (__ballot(qzbits != 0) != 0ull);
   auto patch = [&](vec &v, Real L) { // after a row load: replace the ghost columns by their mirror cells
      if (!VG) return; // (a wave-uniform early-out for segments without ghost columns measured slower: it fences the loads)
      Real lm = lane_from_lower<DPP>(v[V - 1]);
      if (lane == 0) lm = L;
      if (V == 4) {
         if (fix0) v[0] = v[2];
         if (zzN == 1) v[1] = lm;
         if (zzN == 2) v[2] = v[0];
         if (zzN == 3) v[3] = v[1];
      } else {
         const Real up = lane_from_upper<DPP>(v[0]);
         if (fix0) v[0] = up;
         if (zzN == 1) v[1] = lm;
      }
   };
   vec prev[R], cur[R + 2], nxt[R + 2];
   Real curL[R], curR[R], nxtL[R], nxtR[R]; // wave-edge columns (meaningful in lanes 0 / 63 only)
   {
      const Real *pm = u1 + planesrc(xs - 1) * plane;
      const Real *pc = u1 + planesrc(xs) * plane;
#pragma unroll
      for (int r = 0; r < R; r++) prev[r] = *(const vec *)(pm + roff[r + 1]);
#pragma unroll
      for (int j = 0; j < R + 2; j++) cur[j] = *(const vec *)(pc + roff[j]);
#pragma unroll
      for (int r = 0; r < R; r++) {
         curL[r] = need_l ? pc[roff[r + 1] - 1] : Real(0);
         curR[r] = need_r ? pc[roff[r + 1] + V] : Real(0);
         patch(cur[r + 1], curL[r]);
      }
   }
   for (int x = xs; x < xe; x++) {
      const Real *pn = u1 + planesrc(x + 1) * plane;
      Real *po = u0 + (int64_t)x * plane;
      const uint8_t *pmk = mask + (((int64_t)x * plane) >> 3);
      vec old[R];
      uint32_t mb[R];
#pragma unroll
      for (int j = 0; j < R + 2; j++) nxt[j] = *(const vec *)(pn + roff[j]);
#pragma unroll
      for (int r = 0; r < R; r++) {
         old[r] = __builtin_nontemporal_load((const vec *)(po + soff[r]));
         mb[r] = pmk[soff[r] >> 3];
         nxtL[r] = need_l ? pn[roff[r + 1] - 1] : Real(0);
         nxtR[r] = need_r ? pn[roff[r + 1] + V] : Real(0);
         patch(nxt[r + 1], nxtL[r]);
      }
      const bool qx = (VG || ABCK) && ((ap.first && x == 1) || (ap.last && x == ap.Nx - 2));
#pragma unroll
      for (int r = 0; r < R; r++) {
         const vec c = cur[r + 1];
         Real zm = lane_from_lower<DPP>(c[V - 1]);
         Real zp = lane_from_upper<DPP>(c[0]);
         if (lane == 0) zm = curL[r];
         if (lane == LW - 1) zp = curR[r];
         if (fixR) zp = c[V - 2]; // my right neighbour is the ghost column
         const uint32_t bits = mb[r] >> (uint32_t)(soff[r] & 7);
         vec o, lft, rgt;
#pragma unroll
         for (int i = 0; i < V; i++) {
            lft[i] = (i == 0) ? zm : c[i > 0 ? i - 1 : 0];
            rgt[i] = (i == V - 1) ? zp : c[i < V - 1 ? i + 1 : V - 1];
         }
         // file order +x, -x, +y, -y, +z, -z = storage +NzNy, -NzNy, +Nz, -Nz, +1, -1 (axes exchanged: +1, -1, +Nz, -Nz, +NzNy, -NzNy);
         // one wave-uniform branch around the whole row
         if (ap.swz) {
#pragma unroll
            for (int i = 0; i < V; i++) o[i] = upd7<SG>(a1, a2, c[i], old[r][i], rgt[i], lft[i], cur[r + 2][i], cur[r][i], nxt[r + 1][i], prev[r][i]);
         } else {
#pragma unroll
            for (int i = 0; i < V; i++) o[i] = upd7<SG>(a1, a2, c[i], old[r][i], nxt[r + 1][i], prev[r][i], cur[r + 2][i], cur[r][i], rgt[i], lft[i]);
         }
         if (VG || ABCK) { // ABC loss (cpu_engine.h:225-229); u2ba is the old value of the cell
            const int64_t y = y0 + r;
            const int qxy = (qx ? 1 : 0) + ((y == 1 || y == Ny - 2) ? 1 : 0);
            if (qxy > 0 || wave_has_qz) {
#pragma unroll
               for (int i = 0; i < V; i++) {
                  const bool zq = (qzbits >> i) & 1u;
                  if (qxy > 0 || __ballot(zq) != 0ull) {
                     const int Q = qxy + (zq ? 1 : 0);
                     if (Q > 0) {
                        o[i] = abc_loss<SG>(o[i], old[r][i], labc * (Real)Q);
                     }
                  }
               }
            }
         }
#pragma unroll
         for (int i = 0; i < V; i++)
            if ((bits >> i) & 1u) o[i] = old[r][i];
         if (active && (y0 + r <= Ny - 2))
            __builtin_nontemporal_store(o, (vec *)((u0_dst ? u0_dst + (int64_t)x * plane : po) + soff[r]));
      }
#pragma unroll
      for (int r = 0; r < R; r++) {
         prev[r] = cur[r + 1];
         curL[r] = nxtL[r];
         curR[r] = nxtR[r];
      }
#pragma unroll
      for (int j = 0; j < R + 2; j++) cur[j] = nxt[j];
   }
}

// =============================================================================================================
// Air update, 13-point FCC (cpu_engine.h:195-223; gpu_engine.h:245-274), dense form: serves the folded grid
// (fcc_flag 2) directly and the checkerboard grid (fcc_flag 1) through the odd-parity bits of the skip-mask.
// Neighbour order (= accumulation order): (+x+y)(-x-y)(+y+z)(-y-z)(+x+z)(-x-z)(+x-y)(-x+y)(+y-z)(-y+z)(+x-z)(-x+z).
// Same marching scheme; all three planes keep R+2 rows, and every row used with a z offset gets its wave-edge
// columns from the edge lanes.
// =============================================================================================================
template <typename Real, int R, int WY, int WZ, bool SG, bool DPP, bool VG = false, bool ABCK = false, int LW = 64>
__global__ __launch_bounds__(64 * WY * WZ) void k_air_fcc(const Real *__restrict__ u1, Real *u0,
                                                         const uint8_t *__restrict__ mask, Real a1, Real a2,
                                                         AirParams ap, Real labc, const Real *u0_src = nullptr,
                                                         const int32_t *__restrict__ tiles = nullptr) {
   // u0_src: read u^{n-1} there instead of from u0 (out of place: the shell of a temporally blocked pair);
   // tiles: block b works on tile tiles[b] = (xc*nyt + yt)*nzt + zt instead of walking the whole range
   typedef typename VecOf<Real>::type vec;
   constexpr int V = VecOf<Real>::V;
   const uint32_t total = (uint32_t)ap.nzt * ap.nyt * ap.nxc;
   uint32_t b = blockIdx.x;
   if (tiles) b = (uint32_t)tiles[b];
   else if (ap.swizzle == 2) { if (!xcd_band(blockIdx.x, (uint32_t)ap.nzt * ap.nyt, (uint32_t)ap.nxc, b)) return; }
   else if (ap.swizzle) b = xcd_swizzle(b, total);
   const int zt = b % ap.nzt;
   const int yt = (b / ap.nzt) % ap.nyt;
   const int xc = b / (ap.nzt * ap.nyt);
   // LW lanes span a row segment; a wave stacks 64/LW such segments in y (LW < 64: narrow grids, cf. Engine::pick_lw)
   static_assert(LW == 64 || LW == 32 || LW == 16, "row segments are 64, 32 or 16 lanes wide");
   constexpr int NSUB = 64 / LW;
   const int wlane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   const int lane = wlane % LW, sub = wlane / LW; // `lane` = position inside the row segment
   const int wz = wave % WZ, wy = wave / WZ;
   const int64_t z0 = ((int64_t)(zt * WZ + wz) * LW + lane) * V;
   const int64_t y0 = 1 + ((int64_t)(yt * WY + wy) * NSUB + sub) * R;
   const int xs = ap.x_begin + xc * ap.chunk;
   const int xe = min(xs + ap.chunk, ap.x_end);
   const int64_t Ny = ap.Ny, P = ap.P, plane = ap.plane;
   const bool active = z0 < P;
   if (y0 - (int64_t)sub * R > Ny - 2) return;
   const int64_t zl = active ? z0 : 0;
   const bool need_l = (lane == 0) && (z0 > 0);
   const bool need_r = (lane == LW - 1) && (z0 + V < P);

   auto rowsrc = [&](int64_t y) -> int64_t {
      if (y > Ny - 1) y = Ny - 1;
      if (VG) { if (y == 0) return 2; if (y == Ny - 1) return ap.fold ? Ny - 2 : Ny - 3; }
      return y;
   };
   auto planesrc = [&](int x) -> int64_t {
      if (VG) { if (ap.first && x == 0) return 2; if (ap.last && x == ap.Nx - 1) return ap.Nx - 3; }
      return x;
   };
   int64_t roff[R + 2], soff[R];
#pragma unroll
   for (int j = 0; j < R + 2; j++) roff[j] = rowsrc(y0 - 1 + j) * P + zl;
#pragma unroll
   for (int r = 0; r < R; r++) soff[r] = (y0 + r > Ny - 1 ? Ny - 1 : y0 + r) * P + zl;
   const int zzN = VG ? (int)(ap.Nz - 1 - z0) : -1;
   const bool fix0 = VG && (z0 == 0);
   const bool fixR = VG && (zzN == V);
   uint32_t qzbits = 0;
   if (VG || ABCK) {
#pragma unroll
      for (int i = 0; i < V; i++)
         if (active && (z0 + i == 1 || z0 + i == ap.Nz - 2)) qzbits |= 1u << i;
   }
   const bool wave_has_qz = (VG || ABCK) && (__ballot(qzbits != 0) != 0ull);
   auto patch = [&](vec &v, Real L) {
      if (!VG) return; // (a wave-uniform early-out for segments without ghost columns measured slower: it fences the loads)
      Real lm = lane_from_lower<DPP>(v[V - 1]);
      if (lane == 0) lm = L;
      if (V == 4) {
         if (fix0) v[0] = v[2];
         if (zzN == 1) v[1] = lm;
         if (zzN == 2) v[2] = v[0];
         if (zzN == 3) v[3] = v[1];
      } else {
         const Real up = lane_from_upper<DPP>(v[0]);
         if (fix0) v[0] = up;
         if (zzN == 1) v[1] = lm;
      }
   };
   // three planes x (R+2) rows, each with its two wave-edge columns
   vec prev[R + 2], cur[R + 2], nxt[R + 2];
   Real prevL[R + 2], prevR[R + 2], curL[R + 2], curR[R + 2], nxtL[R + 2], nxtR[R + 2];
   {
      const Real *pm = u1 + planesrc(xs - 1) * plane;
      const Real *pc = u1 + planesrc(xs) * plane;
#pragma unroll
      for (int j = 0; j < R + 2; j++) {
         prev[j] = *(const vec *)(pm + roff[j]);
         cur[j] = *(const vec *)(pc + roff[j]);
         prevL[j] = need_l ? pm[roff[j] - 1] : Real(0);
         prevR[j] = need_r ? pm[roff[j] + V] : Real(0);
         curL[j] = need_l ? pc[roff[j] - 1] : Real(0);
         curR[j] = need_r ? pc[roff[j] + V] : Real(0);
         patch(prev[j], prevL[j]);
         patch(cur[j], curL[j]);
      }
   }
   for (int x = xs; x < xe; x++) {
      const Real *pn = u1 + planesrc(x + 1) * plane;
      Real *po = u0 + (int64_t)x * plane;
      const uint8_t *pmk = mask + (((int64_t)x * plane) >> 3);
      vec old[R];
      uint32_t mb[R];
#pragma unroll
      for (int j = 0; j < R + 2; j++) {
         nxt[j] = *(const vec *)(pn + roff[j]);
         nxtL[j] = need_l ? pn[roff[j] - 1] : Real(0);
         nxtR[j] = need_r ? pn[roff[j] + V] : Real(0);
         patch(nxt[j], nxtL[j]);
      }
      const Real *pold = u0_src ? u0_src + (int64_t)x * plane : po;
#pragma unroll
      for (int r = 0; r < R; r++) {
         old[r] = __builtin_nontemporal_load((const vec *)(pold + soff[r]));
         mb[r] = pmk[soff[r] >> 3];
      }
      const bool qx = (VG || ABCK) && ((ap.first && x == 1) || (ap.last && x == ap.Nx - 2));
      // z-shifted views: lo(v)[i] = v[z-1], hi(v)[i] = v[z+1]
      auto shift_lo = [&](const vec &v, Real edge) {
         Real zm = lane_from_lower<DPP>(v[V - 1]);
         if (lane == 0) zm = edge;
         vec s;
#pragma unroll
         for (int i = 0; i < V; i++) s[i] = (i == 0) ? zm : v[i > 0 ? i - 1 : 0];
         return s;
      };
      auto shift_hi = [&](const vec &v, Real edge) {
         Real zp = lane_from_upper<DPP>(v[0]);
         if (lane == LW - 1) zp = edge;
         if (fixR) zp = v[V - 2]; // my right neighbour is the ghost column
         vec s;
#pragma unroll
         for (int i = 0; i < V; i++) s[i] = (i == V - 1) ? zp : v[i < V - 1 ? i + 1 : V - 1];
         return s;
      };
#pragma unroll
      for (int r = 0; r < R; r++) {
         const int j = r + 1; // own row inside the (R+2)-row windows
         const vec c = cur[j];
         const vec cu_lo = shift_lo(cur[j + 1], curL[j + 1]), cu_hi = shift_hi(cur[j + 1], curR[j + 1]);
         const vec cd_lo = shift_lo(cur[j - 1], curL[j - 1]), cd_hi = shift_hi(cur[j - 1], curR[j - 1]);
         const vec n_lo = shift_lo(nxt[j], nxtL[j]), n_hi = shift_hi(nxt[j], nxtR[j]);
         const vec p_lo = shift_lo(prev[j], prevL[j]), p_hi = shift_hi(prev[j], prevR[j]);
         const uint32_t bits = mb[r] >> (uint32_t)(soff[r] & 7);
         vec o;
         // file order of the twelve neighbours = storage +NzNy+Nz, -NzNy-Nz, +Nz+1, -Nz-1, +NzNy+1, -NzNy-1, +NzNy-Nz, -NzNy+Nz, +Nz-1,
         // -Nz+1, +NzNy-1, -NzNy+1; with the axes exchanged file (dx, dy, dz) = storage (dz, dy, dx): entries 0<->2, 1<->3, 6<->9,
         // 7<->8, 10<->11 change places.  One wave-uniform branch around the whole row.
         if (ap.swz) {
#pragma unroll
            for (int i = 0; i < V; i++) {
               const Real nbs[12] = {cu_hi[i], cd_lo[i], nxt[j + 1][i], prev[j - 1][i], n_hi[i], p_lo[i],
                                     cd_hi[i], cu_lo[i], prev[j + 1][i], nxt[j - 1][i], p_hi[i], n_lo[i]};
               o[i] = upd13<SG>(a1, a2, c[i], old[r][i], nbs);
            }
         } else {
#pragma unroll
            for (int i = 0; i < V; i++) {
               const Real nb[12] = {nxt[j + 1][i], prev[j - 1][i], cu_hi[i], cd_lo[i], n_hi[i], p_lo[i],
                                    nxt[j - 1][i], prev[j + 1][i], cu_lo[i], cd_hi[i], n_lo[i], p_hi[i]};
               o[i] = upd13<SG>(a1, a2, c[i], old[r][i], nb);
            }
         }
         if (VG || ABCK) { // ABC loss (cpu_engine.h:225-229)
            const int64_t y = y0 + r;
            const int qxy = (qx ? 1 : 0) + ((y == 1 || (!ap.fold && y == Ny - 2)) ? 1 : 0);
            if (qxy > 0 || wave_has_qz) {
#pragma unroll
               for (int i = 0; i < V; i++) {
                  const bool zq = (qzbits >> i) & 1u;
                  if (qxy > 0 || __ballot(zq) != 0ull) {
                     const int Q = qxy + (zq ? 1 : 0);
                     if (Q > 0) {
                        o[i] = abc_loss<SG>(o[i], old[r][i], labc * (Real)Q);
                     }
                  }
               }
            }
         }
#pragma unroll
         for (int i = 0; i < V; i++)
            if ((bits >> i) & 1u) o[i] = old[r][i];
         if (active && (y0 + r <= Ny - 2)) __builtin_nontemporal_store(o, (vec *)(po + soff[r]));
      }
#pragma unroll
      for (int j = 0; j < R + 2; j++) {
         prev[j] = cur[j]; prevL[j] = curL[j]; prevR[j] = curR[j];
         cur[j] = nxt[j];  curL[j] = nxtL[j];  curR[j] = nxtR[j];
      }
   }
}

// ---- ghost-shell maintenance (cpu_engine.h:135-172; gpu_engine.h:277-285,435-494) ---------------------------
// z faces: one thread per (x,y) row
template <typename Real>
static __global__ void k_flip_z(Real *__restrict__ u1, int64_t nrows, int64_t P, int64_t Nz) {
   const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (r >= nrows) return;
   Real *row = u1 + r * P;
   row[0] = row[2];
   row[Nz - 1] = row[Nz - 3];
}
// y faces (+ the folded-FCC ghost row, which the reference copies before the flips: done by the caller's order)
// mode bit0: row0 <- row2 ; bit1: row Ny-1 <- row Ny-3 ; bit2: row Ny-1 <- row Ny-2 (fold)
template <typename Real>
static __global__ void k_flip_y(Real *__restrict__ u1, int64_t Nx, int64_t Ny, int64_t P, int64_t Nz, int mode) {
   const int64_t iz = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   const int64_t ix = blockIdx.y;
   if (iz >= Nz || ix >= Nx) return;
   Real *pl = u1 + ix * Ny * P + iz;
   if (mode & 4) pl[(Ny - 1) * P] = pl[(Ny - 2) * P];
   if (mode & 1) pl[0] = pl[2 * P];
   if (mode & 2) pl[(Ny - 1) * P] = pl[(Ny - 3) * P];
}
// x faces: plane 0 <- plane 2 (first slab), plane Nx-1 <- plane Nx-3 (last slab)
template <typename Real>
static __global__ void k_flip_x(Real *__restrict__ u1, int64_t Nx, int64_t plane, int first, int last) {
   const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (i >= plane) return;
   if (first) u1[i] = u1[2 * plane + i];
   if (last) u1[(Nx - 1) * plane + i] = u1[(Nx - 3) * plane + i];
}

// ---- ABC (first-order Engquist-Majda), cpu_engine.h:131-134 (save) and :225-229 (loss) ----------------------
template <typename Real>
static __global__ void k_abc_save(const Real *__restrict__ u0, const int64_t *__restrict__ idx, Real *__restrict__ u2ba,
                           int64_t n) {
   const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (i < n) u2ba[i] = u0[idx[i]];
}
template <typename Real, bool SG>
static __global__ void k_abc_loss(Real *__restrict__ u0, const int64_t *__restrict__ idx, const int8_t *__restrict__ Q,
                           const Real *__restrict__ u2ba, Real l, int64_t begin, int64_t end) {
   const int64_t i = begin + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (i >= end) return;
   const int64_t ib = idx[i];
   u0[ib] = abc_loss<SG>(u0[ib], u2ba[i], l * (Real)Q[i]); // (cpu_engine.h:225-229 incl. the double division of :228)
}

// ---- rigid boundary nodes, cpu_engine.h:234-287 (gpu_engine.h:288-348) --------------------------------------
// gather the NN neighbours of a boundary node in the adjacency-bit order (cpu_engine.h:241-246 / 262-273)
// (axes exchanged in storage: sx = stride of the file's x axis = 1, sz = stride of its z axis = plane)
template <typename Real, bool FCC>
// need: bit k set = neighbour k is wanted.  A neighbour whose adjacency bit is clear enters the CPU-exact rigid update as
// p + (a2*0)*u1 = p + (+-0) (upd_rigid), which leaves p as it is unless p is -0 -- and p never is: every sum of the time loop
// has a +0 or a non-zero addend (the state starts as +0, and x + (-x) = +0), so the field holds no -0 and b1*c - old is never
// -0.  Those neighbours (the cells INSIDE the wall) therefore need not be fetched: a third of a floor node's lines.  The
// safeguarded arithmetic (products bit*u1, round-towards-zero tree) passes need = all ones.
__device__ __forceinline__ void gather_nb(const Real *__restrict__ u1, int64_t ii, int64_t P, int64_t plane_, Real (&nb)[FCC ? 12 : 6], bool swz = false,
                                          uint32_t need = 0xffffu) {
   const int64_t plane = swz ? 1 : plane_, one = swz ? plane_ : 1;
   if (!FCC) {
      const int64_t off[6] = {plane, -plane, P, -P, one, -one};
#pragma unroll
      for (int j = 0; j < 6; j++) nb[j] = ((need >> j) & 1u) ? u1[ii + off[j]] : (Real)0;
   } else {
      const int64_t off[12] = {plane + P, -plane - P, P + one, -P - one, plane + one, -plane - one,
                               plane - P, -plane + P, P - one, -P + one, plane - one, -plane + one};
#pragma unroll
      for (int j = 0; j < 12; j++) nb[j] = ((need >> j) & 1u) ? u1[ii + off[j]] : (Real)0;
   }
}
template <typename Real, bool FCC, bool SG>
static __global__ void k_rigid(const Real *__restrict__ u1, Real *__restrict__ u0, const int64_t *__restrict__ idx,
                        const uint16_t *__restrict__ adjv, Real a2, Real sl2, int64_t P, int64_t plane,
                        int64_t begin, int64_t end, int swz) {
   const int64_t nb = begin + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (nb >= end) return;
   const int64_t ii = idx[nb];
   const uint32_t adj = adjv[nb];
   Real v[FCC ? 12 : 6];
   gather_nb<Real, FCC>(u1, ii, P, plane, v, swz != 0);
   u0[ii] = upd_rigid<SG, FCC ? 12 : 6>(a2, sl2, adj, u1[ii], u0[ii], v);
}

// ---- frequency-dependent (lossy) boundary nodes, cpu_engine.h:290-301 + 363-405 (gpu_engine.h:368-432) ------
// gather + ODE update + scatter in one pass; branch states in 64-node blocks (st_idx below) so that lanes coalesce.
template <typename Real> struct MatQuadT { Real b, bd, bDh, bFh; };

// Branch state (vh1, gh1) layout: blocks of 64 nodes, [node / 64][branch m][node % 64].  A wave of 64 consecutive lossy
// nodes then reads and writes ONE contiguous 3 KiB piece per array instead of 256 bytes out of each of 12 streams that lie
// Nbl * sizeof(Real) apart ([m][Nbl], round 1): the list kernels were DRAM-page-bound on those 24 interleaved streams
// (k_fd_sel 3.2 TB/s with the scattered store taken out).  Arrays hold round_up(Nbl, 64) * 12 elements.
__device__ __host__ __forceinline__ int64_t st_idx(int m, int64_t li) { return (((li >> 6) * 12 + m) << 6) + (li & 63); }

// ---- FD (RLC-branch) update of one lossy node, cpu_engine.h:363-405: p = the node's value after the rigid update ------
// (shared by k_boundary and k_fd_boundary, so that all produce the same bits)
// All branch-state and coefficient loads are issued up front (their addresses do not depend on the arithmetic): with the
// loads inside the accumulation loop every branch cost a memory round trip and the list kernels ran latency-bound
// (k_fd_sel 2.6 TB/s); the arithmetic keeps the reference's order.
// vin / gin: the branch state before the step, vout / gout: where the state after it goes (the same arrays for the in-place
// single steps; the other half of a double buffer inside the wall-region pairs, pf_wall.h); wr = false: evaluate only
// (halo nodes of a wall region: their owner stores).  u2 = the node's value two steps back.
template <typename Real>
__device__ __forceinline__ Real fd_core(Real p, Real u2, int32_t li, const Real *vin, const Real *gin, Real *vout, Real *gout, bool wr,
                                        const Real *__restrict__ ssaf, const int8_t *__restrict__ mat, const int8_t *__restrict__ Mb,
                                        const MatQuadT<Real> *__restrict__ mq, const Real *__restrict__ beta, Real lo2, int64_t mmax) {
   // mmax = the largest branch count of any material of the scene (uniform): the branch-state loads are issued for
   // m < mmax right away, without waiting for the node's own count M = Mb[mat[li]] -- two dependent round trips less per
   // wave (the list kernels are latency-bound: index -> material -> count -> state); states m >= M are loaded and ignored
   const Real two = 2.0, one = 1.0;
   Real v1[12], g1[12];
#pragma unroll
   for (int m = 0; m < 12; m++) {
      if (m < (int)mmax) {
         v1[m] = vin[st_idx(m, li)]; // (plain, not nontemporal, loads and stores: measured 12 % faster for k_fd_sel, 5 % for k_boundary)
         g1[m] = gin[st_idx(m, li)];
      }
   }
   const int32_t k = mat[li];
   const int M = Mb[k];
   MatQuadT<Real> q[12];
#pragma unroll
   for (int m = 0; m < 12; m++)
      if (m < (int)mmax) q[m] = mq[k * 12 + m];
   const Real sf = ssaf[li];
   const Real g = lo2 * sf * beta[k];
   const Real fac = two * lo2 * sf / (one + g);
   Real u = p;
   u = (u + g * u2) / (one + g);
#pragma unroll
   for (int m = 0; m < 12; m++)
      if (m < M) u -= fac * (two * q[m].bDh * v1[m] - q[m].bFh * g1[m]);
   const Real du = u - u2;
#pragma unroll
   for (int m = 0; m < 12; m++) {
      if (m < M) {
         const Real v0 = q[m].b * du + q[m].bd * v1[m] - two * q[m].bFh * g1[m];
         if (wr) {
            gout[st_idx(m, li)] = g1[m] + (v0 + v1[m]) / two;
            vout[st_idx(m, li)] = v0;
         }
      }
   }
   return u;
}
template <typename Real>
__device__ __forceinline__ Real fd_node_update(Real p, int32_t li, Real *u0b, const Real *u2b,
                                               const Real *__restrict__ ssaf, const int8_t *__restrict__ mat,
                                               const int8_t *__restrict__ Mb, const MatQuadT<Real> *__restrict__ mq,
                                               const Real *__restrict__ beta, const Real *vh1, const Real *gh1, Real *vh1o, Real *gh1o,
                                               Real lo2, int64_t mmax) {
   const Real u = fd_core<Real>(p, u2b[li], li, vh1, gh1, vh1o, gh1o, true, ssaf, mat, Mb, mq, beta, lo2, mmax);
   u0b[li] = u;
   return u;
}

// ---- frequency-dependent (lossy) boundary nodes as a separate pass, cpu_engine.h:290-301 + 363-405 (gpu_engine.h:368-432):
// gather + ODE update + scatter; branch states in 64-node blocks [node / 64][branch][node % 64] (st_idx).
template <typename Real>
static __global__ void k_fd_boundary(Real *__restrict__ u0, const int64_t *__restrict__ idx, Real *__restrict__ u0b,
                              const Real *__restrict__ u2b, const Real *__restrict__ ssaf,
                              const int8_t *__restrict__ mat, const int8_t *__restrict__ Mb,
                              const MatQuadT<Real> *__restrict__ mq, const Real *__restrict__ beta,
                              Real *vh1, Real *gh1, Real lo2, int64_t mmax, int64_t begin,
                              int64_t end) {
   const int64_t nb = begin + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (nb >= end) return;
   const int64_t ii = idx[nb];
   u0[ii] = fd_node_update<Real>(u0[ii], (int32_t)nb, u0b, u2b, ssaf, mat, Mb, mq, beta, vh1, gh1, vh1, gh1, lo2, mmax);
}

// ---- fused boundary pass: rigid update of every boundary node + FD update of the lossy ones in one visit ----------
// (cpu_engine.h:234-287 then :290-301,363-405 for the same node: identical arithmetic, the gather / scatter of u0
// between the two is a register).  lossy[nb] = index into the lossy-node arrays, or -1 for a rigid node; lossy
// indices increase along the (sorted) boundary list, so the branch-state accesses stay coalesced.
// sel != null: visit the nodes sel[begin..end) instead of begin..end (temporal blocking leaves the column-strip nodes
// to k_air_zstrip).
template <typename Real, bool FCC, bool SG>
static __global__ void k_boundary(const Real *__restrict__ u1, Real *u0, const int64_t *__restrict__ idx,
                           const uint16_t *__restrict__ adjv, const int32_t *__restrict__ lossy, Real a2, Real sl2,
                           int64_t P, int64_t plane, Real *u0b, const Real *u2b, // (u0b may be u2b: second step of a wall-region pair)
                           const Real *__restrict__ ssaf, const int8_t *__restrict__ mat, const int8_t *__restrict__ Mb,
                           const MatQuadT<Real> *__restrict__ mq, const Real *__restrict__ beta, const Real *vh1,
                           const Real *gh1, Real *vh1o, Real *gh1o, // branch state before / after (the same arrays: in place)
                           Real lo2, int64_t mmax, int64_t begin, int64_t end,
                           const Real *u0_old, const int32_t *__restrict__ sel, int swz, // u0_old: where u^{n-1} lives (== u0 in place)
                           const int32_t *__restrict__ fdsel = nullptr, int64_t nfd = 0, const int64_t *__restrict__ idx_l = nullptr,
                           int flags = 0) {
   // flags & 1: XCD-aware order of the workgroups over the list.  G = flags >> 4 > 0: inside every window of 8*G workgroups, XCD k
   // (blocks k, k+8, ...) walks the k-th run of G consecutive pieces, so the rows of u^n a node row shares with the next one (all
   // but the outermost of its 3 / 9 neighbour rows) are asked for by the same L2 a few workgroups later instead of by up to
   // eight L2s, while all XCDs stay within 8*G pieces of each other (G = 0: one run per XCD over the whole list).
   // flags & 2: fetch every neighbour (gather_nb).
   // fdsel: threads beyond the list do the branch ODEs of nfd lossy nodes whose RIGID update was done elsewhere (the column-strip
   // kernel left it in u0b[li]): k_fd_sel's work in the same launch -- one kernel, one wait for the index chains, fewer
   uint32_t blk = blockIdx.x;
   if (flags & 1) {
      const uint32_t G = (uint32_t)flags >> 4; // run length per XCD (workgroups); 0: one run per XCD over the whole list
      if (G == 0) blk = xcd_swizzle(blockIdx.x, gridDim.x);
      else {
         const uint32_t win = 8u * G, full = (gridDim.x / win) * win;
         if (blk < full) { const uint32_t r = blk % win; blk = blk - r + (r & 7u) * G + (r >> 3); }
      }
   }
   const int64_t t = begin + blk * (int64_t)blockDim.x + threadIdx.x;
   if (t >= end) {
      const int64_t f = t - end;
      if (f < nfd) {
         const int32_t li = fdsel[f];
         u0[idx_l[li]] = fd_node_update<Real>(u0b[li], li, u0b, u2b, ssaf, mat, Mb, mq, beta, vh1, gh1, vh1o, gh1o, lo2, mmax);
      }
      return;
   }
   const int64_t nb = sel ? (int64_t)sel[t] : t;
   const int64_t ii = idx[nb];
   const uint32_t adj = adjv[nb];
   Real v[FCC ? 12 : 6];
   gather_nb<Real, FCC>(u1, ii, P, plane, v, swz != 0, (SG || (flags & 2)) ? 0xffffu : adj);
   Real p = upd_rigid<SG, FCC ? 12 : 6>(a2, sl2, adj, u1[ii], u0_old[ii], v);
   const int32_t li = lossy[nb];
   if (li >= 0) p = fd_node_update<Real>(p, li, u0b, u2b, ssaf, mat, Mb, mq, beta, vh1, gh1, vh1o, gh1o, lo2, mmax);
   u0[ii] = p;
}

// device-side step counters of the graph-replayed loop: ctr[0] = step index, ctr[1] = receiver ring column
static __global__ void k_ctr_set(int64_t *ctr, int64_t n, int64_t col) { ctr[0] = n; ctr[1] = col; }
static __global__ void k_ctr_tick(int64_t *ctr) { ctr[0]++; ctr[1]++; }

// ---- receivers (read time n from u1) and sources (add to time n+1 in u0), cpu_engine.h:304-313 --------------
// one launch: threads [0,Nr) gather into the ring column, thread Nr (alone) applies all sources in list order
// (the reference loop is serial, so duplicate source nodes accumulate in order).
template <typename Real>
static __global__ void k_io(const Real *__restrict__ u1, Real *__restrict__ u0, const int64_t *__restrict__ out_idx,
                     Real *__restrict__ ring, int64_t Nr, int64_t ring_col, int64_t ring_depth,
                     const int64_t *__restrict__ in_idx, const Real *__restrict__ in_sigs, int64_t Ns, int64_t Nt,
                     int64_t n, const int64_t *__restrict__ ctr = nullptr, const Real *__restrict__ u1b = nullptr) {
   if (ctr) { n = ctr[0]; ring_col = ctr[1]; } // replayed from a hipGraph: step index and ring column live on the device
   const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t < Nr) {
      ring[t * ring_depth + ring_col] = u1[out_idx[t]];
      if (u1b) ring[t * ring_depth + ring_col + 1] = u1b[out_idx[t]]; // (the next step's readout in the same launch: Engine::step_triple)
   } else if (t == Nr) {
      for (int64_t s = 0; s < Ns; s++) u0[in_idx[s]] += in_sigs[s * Nt + n];
   }
}

// build the engine's skip-mask rows for ghost z columns / pad / odd parity (boundary-node bits are OR-ed in
// afterwards by k_mask_set)
static __global__ void k_mask_init(uint8_t *__restrict__ mask, int64_t Nx, int64_t Ny, int64_t P, int64_t Nz, int parity) {
   // one thread per 32 cells (one mask word) of a row: P is a multiple of 32 for fp32 and of 16 for fp64 (two bytes then)
   const int64_t wpr = P / 16;                           // 16-bit pieces per row
   const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t >= Nx * Ny * wpr) return;
   const int64_t row = t / wpr, piece = t % wpr;
   const int64_t iy = row % Ny, ix = row / Ny;
   const int64_t iz0 = piece * 16;
   uint32_t m = 0;
   for (int i = 0; i < 16; i++) {
      const int64_t iz = iz0 + i;
      bool skip = (iz == 0) || (iz >= Nz - 1);
      if (parity && (((ix + iy + iz + (parity - 1)) & 1) != 0)) skip = true; // parity-1 = parity of the global ix of plane 0
      if (skip) m |= 1u << i;
   }
   ((uint16_t *)mask)[row * wpr + piece] = (uint16_t)m;
}
// exchanged-axes storage (Engine::swz) <-> file order: file cell (fx, fy, fz) lives at storage ((fz*Ny + fy)*P + fx)
template <typename Real>
static __global__ void k_storage_to_file(const Real *__restrict__ st, Real *__restrict__ file, int64_t fNx, int64_t fNy, int64_t fNz, int64_t Ny, int64_t P) {
   const int64_t ii = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (ii >= fNx * fNy * fNz) return;
   const int64_t fz = ii % fNz, fy = (ii / fNz) % fNy, fx = ii / (fNz * fNy);
   file[ii] = st[(fz * Ny + fy) * P + fx];
}
template <typename Real>
static __global__ void k_file_to_storage(const Real *__restrict__ file, Real *__restrict__ st, int64_t fNx, int64_t fNy, int64_t fNz, int64_t Ny, int64_t P) {
   const int64_t ii = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (ii >= fNx * fNy * fNz) return;
   const int64_t fz = ii % fNz, fy = (ii / fNz) % fNy, fx = ii / (fNz * fNy);
   st[(fz * Ny + fy) * P + fx] = file[ii];
}
static __global__ void k_mask_set(uint8_t *__restrict__ mask, const int64_t *__restrict__ idx, int64_t n) {
   const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (i >= n) return;
   const int64_t jj = idx[i];
   // byte-granular atomic OR through the containing 32-bit word
   uint32_t *w = (uint32_t *)(mask + ((jj >> 3) & ~(int64_t)3));
   atomicOr(w, (1u << (jj & 7)) << (8 * ((jj >> 3) & 3)));
}

} // namespace pf
