// pf_tb2.h -- temporal blocking: k_tb2_reg (two leap-frog steps of the pure 7-point air update per pass over a
// boundary-free region) and k_air_zstrip (single-step update of the thin column strips beside it).  Both are on the
// product path (Engine::step_pair).  Out of place: reads A = u^{n-1}, B = u^n, writes C = u^{n+1}, D = u^{n+2}.
// No mask / ABC / boundary nodes: valid for cells at least 3 away from anything special.
// (The earlier LDS-based research prototypes live in tools/csrc/pf_probe_kernels.h, outside the product library.)
#pragma once
#include "pf_kernels.h"

namespace pf {

struct Tb2Params {
   const void *A, *B;             // float for the prototypes, Real for k_tb2_reg / k_tb2_lds
   void *C, *D;
   int64_t plane;
   int32_t Nx, Ny, Nz, P;
   int32_t x_begin, x_end, chunk; // D planes [x_begin, x_end) (C planes x_begin .. x_end)
   int32_t nzt, nyt, nxc;
   int32_t y_begin, z_begin;      // first core row / first core column of tile (0,0)
   int32_t y_end, z_end;          // one past the last core row / column (0: Ny - y_begin / Nz - z_begin)
   int32_t band;                  // 1: XCD k (blocks k, k+8, ...) works on a contiguous band of y-z tiles of every x chunk
   const int32_t *tiles;          // non-null: block b works on tile tiles[b] = (xc*nyt + yt)*nzt + zt (rooms with interior
                                  // geometry: the clean tiles for k_tb2_reg, the others for k_tb1_tile)
   const uint8_t *mask;           // k_tb1_tile: the engine's skip-mask
   int32_t xsub;                  // k_tb1_tile: > 1: that many workgroups share a tile, each marching a piece of its x chunk
   void *E;                       // k_tb3 (pf_tb3.h): u^{n+3}; C = scratch for the u^{n+1} of flagged tiles, D = u^{n+2}
   // k_tb3<..., SRC = true> (round 6): the sources, added in the kernel after every stage (cpu_engine.h:303-306: u0[in_ixyz] += in_sigs[n])
   const int64_t *src_idx;        // cells (storage order), list order
   const void *src_sig;           // Real [nsrc][src_Nt]
   int64_t src_Nt, src_n;         // samples per source; the step stage 1 computes (stage s adds sample src_n + s - 1)
   int32_t nsrc;
};

// ---------------------------------------------------------------------------------------------------------------
// k_tb2_reg -- the same two-steps-per-pass scheme with every operand in registers (no LDS, no barriers).
// A wave owns 256 columns x R rows of the u^{n+2} output and marches x.  Lanes 0 and 63 are z halo (their u^{n+1}
// values feed lanes 1 / 62 through the DPP wave shifts), so tiles overlap by 8 columns; in y every lane computes
// u^{n+1} on its R rows + 1 above + 1 below from u^n rows R+4 (halo rows re-read through L1/L2 by the waves above and
// below).  Per plane and lane: R+4 row loads of u^n, R+2 of u^{n-1}, R stores of u^{n+1}, R of u^{n+2}.
// ---------------------------------------------------------------------------------------------------------------
// PROBE: the same code under another name, for the creation-time measurements (grid placement search, path choice): per-kernel
// profiler statistics of k_tb2_reg<..., false> then hold the launches of the time loop only.
// SG: the reference GPU engine's arithmetic (pf_kernels.h: upd7<true>) instead of the C CPU engine's.
// SWZ: the grid is stored with the file's x and z axes exchanged (Engine::swz): the neighbours enter the sum in the FILE's order
// +x, -x, +y, -y, +z, -z = storage +z, -z, +y, -y, +x, -x.
template <typename Real, int R, int WY, bool NTA = true, int LW = 64, bool PROBE = false, bool SG = false, bool SWZ = false>
__global__ __launch_bounds__(64 * WY) void k_tb2_reg(Tb2Params tp, Real a1, Real a2) {
   typedef typename VecOf<Real>::type vec;
   // LW lanes span a row segment of LW*V columns whose first and last lane are z halo (their u^{n+1} values feed their
   // neighbours through the DPP wave shifts; what the shifts carry across a segment border only ever lands in a halo
   // lane's outer columns, which nobody reads); a wave stacks 64/LW such segments in y (narrow grids, cf. Engine::pick_lw)
   static_assert(LW == 64 || LW == 32 || LW == 16, "row segments are 64, 32 or 16 lanes wide");
   constexpr int V = VecOf<Real>::V, W = LW * V, NSUB = 64 / LW; // columns per lane / per segment (256 fp32, 128 fp64 at LW=64)
   // plain order by default: the XCD swizzle of the single-step kernels costs 12 % here (measured).  band: all XCDs
   // stay on the same x chunk, but each takes a contiguous band of its y-z tiles, so tiles that share halo rows share an L2.
   uint32_t b = blockIdx.x;
   int zt, yt, xc;
   if (tp.tiles) {
      const uint32_t t = (uint32_t)tp.tiles[b];
      zt = t % tp.nzt; yt = (t / tp.nzt) % tp.nyt; xc = t / (tp.nzt * tp.nyt);
   } else if (tp.band) {
      const uint32_t T = (uint32_t)tp.nzt * tp.nyt, Tp = (T + 7) / 8;
      xc = b / (8 * Tp);
      const uint32_t r = b % (8 * Tp), j = (r % 8) * Tp + r / 8;
      if (j >= T || xc >= tp.nxc) return;
      zt = j % tp.nzt; yt = j / tp.nzt;
   } else {
      zt = b % tp.nzt; yt = (b / tp.nzt) % tp.nyt; xc = b / (tp.nzt * tp.nyt);
   }
   const int wlane = threadIdx.x & 63, w = threadIdx.x >> 6;
   const int lane = wlane % LW, sub = wlane / LW;
   const int ze0 = tp.z_begin - V + zt * (W - 2 * V);
   const int yo = tp.y_begin + ((yt * WY + w) * NSUB + sub) * R; // first output row of this lane's segment
   const int xs = tp.x_begin + xc * tp.chunk, xe = min(xs + tp.chunk, tp.x_end);
   const int P = tp.P;
   const int64_t plane = tp.plane;
   const int zc = min(max(ze0 + lane * V, 0), P - V);
   int64_t offB[R + 4];                                       // rows yo-2 .. yo+R+1
#pragma unroll
   for (int i = 0; i < R + 4; i++) offB[i] = (int64_t)min(max(yo - 2 + i, 0), tp.Ny - 1) * P + zc;
   const int z_end = tp.z_end ? tp.z_end : tp.Nz - tp.z_begin, y_end = tp.y_end ? tp.y_end : tp.Ny - tp.y_begin;
   const bool core_col = (lane >= 1 && lane <= LW - 2) && (ze0 + lane * V + V - 1 < z_end);
   bool core_row[R];
#pragma unroll
   for (int r = 0; r < R; r++) core_row[r] = (yo + r < y_end);

   auto loadB = [&](int x, vec *d) {
      const Real *pl = (const Real *)tp.B + (int64_t)x * plane;
#pragma unroll
      for (int i = 0; i < R + 4; i++) d[i] = *(const vec *)(pl + offB[i]);
   };
   auto loadA = [&](int x, vec *d) { // rows yo-1 .. yo+R
      const Real *pl = (const Real *)tp.A + (int64_t)x * plane;
#pragma unroll
      for (int j = 0; j < R + 2; j++) d[j] = NTA ? __builtin_nontemporal_load((const vec *)(pl + offB[j + 1])) : *(const vec *)(pl + offB[j + 1]);
   };
   auto stencil = [&](const vec &c, const vec &xp, const vec &xm, const vec &yp, const vec &ym, const vec &old) {
      const Real lf = lane_from_lower<true>(c[V - 1]);
      const Real rt = lane_from_upper<true>(c[0]);
      vec o;
#pragma unroll
      for (int i = 0; i < V; i++) {
         const Real zp = (i == V - 1) ? rt : c[i < V - 1 ? i + 1 : V - 1];
         const Real zm = (i == 0) ? lf : c[i > 0 ? i - 1 : 0];
         if constexpr (SWZ) o[i] = upd7<SG>(a1, a2, c[i], old[i], zp, zm, yp[i], ym[i], xp[i], xm[i]);
         else o[i] = upd7<SG>(a1, a2, c[i], old[i], xp[i], xm[i], yp[i], ym[i], zp, zm);
      }
      return o;
   };

   vec Bp[R + 2], Bc[R + 4], Bn[R + 4], Bnn[R + 4], Ar[R + 2], Arn[R + 2];
   vec vm[R], vc[R + 2], vn[R + 2];
   {
      vec t[R + 4];
      loadB(xs - 2, t);
#pragma unroll
      for (int j = 0; j < R + 2; j++) Bp[j] = t[j + 1];
      loadB(xs - 1, Bc);
      loadB(xs, Bn);
      loadA(xs - 1, Ar);
   }
#pragma unroll
   for (int r = 0; r < R; r++) vm[r] = vec{};
#pragma unroll
   for (int j = 0; j < R + 2; j++) vc[j] = vec{};
   for (int x1 = xs - 1; x1 <= xe; x1++) {                    // x1: plane of the u^{n+1} values computed this turn
      if (x1 < xe) { loadB(x1 + 2, Bnn); loadA(x1 + 1, Arn); }
      // stage 1: u^{n+1}(x1) on rows yo-1 .. yo+R
#pragma unroll
      for (int j = 0; j < R + 2; j++) vn[j] = stencil(Bc[j + 1], Bn[j + 1], Bp[j], Bc[j + 2], Bc[j], Ar[j]);
      if (x1 >= xs && x1 < xe) {
         Real *pc = (Real *)tp.C + (int64_t)x1 * plane;
#pragma unroll
         for (int r = 0; r < R; r++)
            if (core_col && core_row[r]) __builtin_nontemporal_store(vn[r + 1], (vec *)(pc + offB[r + 2]));
      }
      // stage 2: u^{n+2}(x1-1) from u^{n+1} planes x1-2 (vm), x1-1 (vc), x1 (vn) and u^n plane x1-1 (Bp)
      if (x1 - 1 >= xs) {
         Real *pd = (Real *)tp.D + (int64_t)(x1 - 1) * plane;
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
   }
}

// S = the twelve neighbours in storage terms, in the order a file-order grid accumulates them:
//   0 (+x+y) 1 (-x-y) 2 (+y+z) 3 (-y-z) 4 (+x+z) 5 (-x-z) 6 (+x-y) 7 (-x+y) 8 (+y-z) 9 (-y+z) 10 (+x-z) 11 (-x+z);
// with the axes exchanged the file's k-th neighbour is S[perm[k]] (entries 0<->2, 1<->3, 6<->9, 7<->8, 10<->11 change places)
template <bool SWZ> struct FccOrder;
template <> struct FccOrder<false> { static constexpr int p[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11}; };
template <> struct FccOrder<true> { static constexpr int p[12] = {2, 3, 0, 1, 4, 5, 9, 8, 7, 6, 11, 10}; };

// ---------------------------------------------------------------------------------------------------------------
// k_tb2_fcc -- the two-steps-per-pass scheme for the 13-point FCC stencil on the folded grid (cpu_engine.h:195-223;
// neighbour / accumulation order (+x+y)(-x-y)(+y+z)(-y-z)(+x+z)(-x-z)(+x-y)(-x+y)(+y-z)(-y+z)(+x-z)(-x+z) as k_air_fcc).
// Same tiling as k_tb2_reg; every neighbour lies on a diagonal, so all three u^n planes of a turn need their rows above and
// below (R+4 rows each) and all three u^{n+1} planes R+2 rows.  R = 2: 44 16-byte vectors of state per lane.
// The z +-1 neighbours of a row are the row itself shifted by one column: in-lane for three of four columns, one DPP
// wave shift for the fourth.
// ---------------------------------------------------------------------------------------------------------------
template <typename Real, int R, int WY, int LW = 64, bool SG = false, bool SWZ = false>
__global__ __launch_bounds__(64 * WY) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_tb2_fcc(Tb2Params tp, Real a1, Real a2_) {
   typedef typename VecOf<Real>::type vec;
   static_assert(LW == 64 || LW == 32 || LW == 16, "row segments are 64, 32 or 16 lanes wide");
   constexpr int V = VecOf<Real>::V, W = LW * V, NSUB = 64 / LW;
   const uint32_t t = tp.tiles ? (uint32_t)tp.tiles[blockIdx.x] : blockIdx.x;
   const int zt = t % tp.nzt, yt = (t / tp.nzt) % tp.nyt, xc = t / (tp.nzt * tp.nyt);
   const int wlane = threadIdx.x & 63, w = threadIdx.x >> 6;
   const int lane = wlane % LW, sub = wlane / LW;
   const int ze0 = tp.z_begin - V + zt * (W - 2 * V);
   const int yo = tp.y_begin + ((yt * WY + w) * NSUB + sub) * R;
   const int xs = tp.x_begin + xc * tp.chunk, xe = min(xs + tp.chunk, tp.x_end);
   const int P = tp.P;
   const int64_t plane = tp.plane;
   const int zc = min(max(ze0 + lane * V, 0), P - V);
   uint32_t offB[R + 4];                                      // rows yo-2 .. yo+R+1 (a plane holds < 2^31 elements)
#pragma unroll
   for (int i = 0; i < R + 4; i++) offB[i] = (uint32_t)min(max(yo - 2 + i, 0), tp.Ny - 1) * (uint32_t)P + (uint32_t)zc;
   const int z_end = tp.z_end ? tp.z_end : tp.Nz - tp.z_begin, y_end = tp.y_end ? tp.y_end : tp.Ny - tp.y_begin;
   const bool core_col = (lane >= 1 && lane <= LW - 2) && (ze0 + lane * V + V - 1 < z_end);
   bool core_row[R];
#pragma unroll
   for (int r = 0; r < R; r++) core_row[r] = (yo + r < y_end);

   auto loadB = [&](int x, vec *d) {
      const Real *pl = (const Real *)tp.B + (int64_t)x * plane;
#pragma unroll
      for (int i = 0; i < R + 4; i++) d[i] = *(const vec *)(pl + offB[i]);
   };
   auto loadA = [&](int x, vec *d) { // rows yo-1 .. yo+R
      const Real *pl = (const Real *)tp.A + (int64_t)x * plane;
#pragma unroll
      for (int j = 0; j < R + 2; j++) d[j] = __builtin_nontemporal_load((const vec *)(pl + offB[j + 1]));
   };
   // c: the row itself (plane x); cU / cD: rows y+1 / y-1 of plane x; n*: plane x+1, p*: plane x-1 (U, C, D = rows y+1, y, y-1)
   auto stencil = [&](const vec &c, const vec &old, const vec &cU, const vec &cD, const vec &nU, const vec &nC, const vec &nD,
                      const vec &pU, const vec &pC, const vec &pD) {
      const Real cUm = lane_from_lower<true>(cU[V - 1]), cUp = lane_from_upper<true>(cU[0]);
      const Real cDm = lane_from_lower<true>(cD[V - 1]), cDp = lane_from_upper<true>(cD[0]);
      const Real nCm = lane_from_lower<true>(nC[V - 1]), nCp = lane_from_upper<true>(nC[0]);
      const Real pCm = lane_from_lower<true>(pC[V - 1]), pCp = lane_from_upper<true>(pC[0]);
      // every row multiplies with its own, opaque copy of a2: otherwise the compiler shares the products a2 * u between
      // the rows that use the same cell, keeps them all alive and spills (348 registers wanted, 256 to be had)
      Real a2 = a2_;
      asm volatile("" : "+s"(a2));
      vec o;
#pragma unroll
      for (int i = 0; i < V; i++) {
         const int im = i > 0 ? i - 1 : 0, ip = i < V - 1 ? i + 1 : V - 1;
         // storage order (+x+y)(-x-y)(+y+z)(-y-z)(+x+z)(-x-z)(+x-y)(-x+y)(+y-z)(-y+z)(+x-z)(-x+z); accumulated in the FILE's order (FccOrder)
         const Real S[12] = {nU[i], pD[i], (i == V - 1) ? cUp : cU[ip], (i == 0) ? cDm : cD[im],
                             (i == V - 1) ? nCp : nC[ip], (i == 0) ? pCm : pC[im], nD[i], pU[i],
                             (i == 0) ? cUm : cU[im], (i == V - 1) ? cDp : cD[ip], (i == 0) ? nCm : nC[im], (i == V - 1) ? pCp : pC[ip]};
         if constexpr (SG) {
            Real nb[12];
#pragma unroll
            for (int k = 0; k < 12; k++) nb[k] = S[FccOrder<SWZ>::p[k]];
            o[i] = upd13<true>(a1, a2, c[i], old[i], nb);
         } else {
            Real p = a1 * c[i] - old[i];
#pragma unroll
            for (int k = 0; k < 12; k++) p = p + a2 * S[FccOrder<SWZ>::p[k]];
            o[i] = p;
         }
      }
      return o;
   };

   // Three u^n plane buffers and three u^{n+1} plane buffers whose roles (x-1, x, x+1) rotate with the turn; the x loop is
   // unrolled by three so that the rotation is a renaming, not register moves (a first version with a fourth "prefetch"
   // buffer and moves spilled and ran 2.7x slower than the single-step kernel).  The plane x1+2 is loaded into the buffer
   // of plane x1-1 as soon as stage 1 is done with it (its R centre rows -- the old values of stage 2 -- are set aside),
   // stage 2 and the other wave of the SIMD cover the latency.
   vec b0[R + 4], b1[R + 4], b2[R + 4], Ar[R + 2], Bold[R];
   vec v0[R + 2], v1[R + 2], v2[R + 2];
   loadB(xs - 2, b0);
   loadB(xs - 1, b1);
   loadB(xs, b2);
   loadA(xs - 1, Ar);
#pragma unroll
   for (int j = 0; j < R + 2; j++) { v0[j] = vec{}; v1[j] = vec{}; }
   // one turn: x1 = plane of the u^{n+1} values computed; Pm / Pc / Pn = u^n planes x1-1 / x1 / x1+1; Vm / Vc = u^{n+1}
   // planes x1-2 / x1-1, Vn receives plane x1
   auto turn = [&](int x1, vec(&Pm)[R + 4], vec(&Pc)[R + 4], vec(&Pn)[R + 4], vec(&Vm)[R + 2], vec(&Vc)[R + 2], vec(&Vn)[R + 2]) {
      // stage 1: u^{n+1}(x1) on rows yo-1 .. yo+R
#pragma unroll
      for (int j = 0; j < R + 2; j++) {
         Vn[j] = stencil(Pc[j + 1], Ar[j], Pc[j + 2], Pc[j], Pn[j + 2], Pn[j + 1], Pn[j], Pm[j + 2], Pm[j + 1], Pm[j]);
         __builtin_amdgcn_sched_barrier(0); // row by row: interleaving the rows for ILP costs more registers than there are
      }
#pragma unroll
      for (int r = 0; r < R; r++) Bold[r] = Pm[r + 2];
      if (x1 < xe) { loadB(x1 + 2, Pm); loadA(x1 + 1, Ar); }
      if (x1 >= xs && x1 < xe) {
         Real *pc = (Real *)tp.C + (int64_t)x1 * plane;
#pragma unroll
         for (int r = 0; r < R; r++)
            if (core_col && core_row[r]) __builtin_nontemporal_store(Vn[r + 1], (vec *)(pc + offB[r + 2]));
      }
      // stage 2: u^{n+2}(x1-1) from u^{n+1} planes x1-2, x1-1, x1; its old value is u^n(x1-1)
      if (x1 - 1 >= xs) {
         Real *pd = (Real *)tp.D + (int64_t)(x1 - 1) * plane;
#pragma unroll
         for (int r = 0; r < R; r++) {
            const vec o = stencil(Vc[r + 1], Bold[r], Vc[r + 2], Vc[r], Vn[r + 2], Vn[r + 1], Vn[r], Vm[r + 2], Vm[r + 1], Vm[r]);
            if (core_col && core_row[r]) __builtin_nontemporal_store(o, (vec *)(pd + offB[r + 2]));
            __builtin_amdgcn_sched_barrier(0);
         }
      }
   };
   for (int x1 = xs - 1; x1 <= xe; x1 += 3) {
      turn(x1, b0, b1, b2, v0, v1, v2);
      if (x1 + 1 > xe) break;
      turn(x1 + 1, b1, b2, b0, v1, v2, v0);
      if (x1 + 2 > xe) break;
      turn(x1 + 2, b2, b0, b1, v2, v0, v1);
   }
}

// ---------------------------------------------------------------------------------------------------------------
// k_tb2_fcc_x -- k_tb2_fcc with the u^{n+1} halo rows EXCHANGED between the waves of a workgroup through LDS instead of
// recomputed.  The 13-point kernel is VALU-bound in the bit-exact mode, and with R = 2 the recomputation doubles stage 1
// (4 row updates for 2 useful).  Here a workgroup is WT waves stacked in y, each computing stage 1 on its own R rows only;
// the first and the last wave are pure halo providers (no stage 2, no stores), the WT-2 waves between them own the tile's
// (WT-2)*R rows.  Per plane: every wave publishes its R new u^{n+1} rows to a double-buffered LDS tile, one barrier, the
// inner waves pick up the row above and the row below.  Row updates per plane and workgroup: WT*R + (WT-2)*R for
// (WT-2)*R*2 useful (WT = 8: 1.17x instead of 1.5x), u^n rows loaded per wave R+2 instead of R+4, u^{n-1} rows R instead
// of R+2, and 28 instead of 40 vectors of state per lane.  64-lane row segments only.
// ---------------------------------------------------------------------------------------------------------------
template <typename Real, int R, int WT>
__global__ __launch_bounds__(64 * WT) void k_tb2_fcc_x(Tb2Params tp, Real a1, Real a2_) {
   typedef typename VecOf<Real>::type vec;
   constexpr int V = VecOf<Real>::V, W = 64 * V;
   __shared__ __attribute__((aligned(16))) Real sV[2][WT * R][W];
   const uint32_t t = tp.tiles ? (uint32_t)tp.tiles[blockIdx.x] : blockIdx.x;
   const int zt = t % tp.nzt, yt = (t / tp.nzt) % tp.nyt, xc = t / (tp.nzt * tp.nyt);
   const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
   const bool inner = w >= 1 && w <= WT - 2;
   const int ze0 = tp.z_begin - V + zt * (W - 2 * V);
   const int yo = tp.y_begin + (yt * (WT - 2) + (w - 1)) * R;   // first own row of this wave (wave 0: the R rows above the tile)
   const int xs = tp.x_begin + xc * tp.chunk, xe = min(xs + tp.chunk, tp.x_end);
   const int P = tp.P;
   const int64_t plane = tp.plane;
   const int zc = min(max(ze0 + lane * V, 0), P - V);
   uint32_t offB[R + 2];                                      // rows yo-1 .. yo+R
#pragma unroll
   for (int i = 0; i < R + 2; i++) offB[i] = (uint32_t)min(max(yo - 1 + i, 0), tp.Ny - 1) * (uint32_t)P + (uint32_t)zc;
   const int z_end = tp.z_end ? tp.z_end : tp.Nz - tp.z_begin, y_end = tp.y_end ? tp.y_end : tp.Ny - tp.y_begin;
   const bool core_col = inner && (lane >= 1 && lane <= 62) && (ze0 + lane * V + V - 1 < z_end);
   bool core_row[R];
#pragma unroll
   for (int r = 0; r < R; r++) core_row[r] = (yo + r < y_end);

   auto loadB = [&](int x, vec *d) {
      const Real *pl = (const Real *)tp.B + (int64_t)x * plane;
#pragma unroll
      for (int i = 0; i < R + 2; i++) d[i] = *(const vec *)(pl + offB[i]);
   };
   auto loadA = [&](int x, vec *d) { // rows yo .. yo+R-1
      const Real *pl = (const Real *)tp.A + (int64_t)x * plane;
#pragma unroll
      for (int j = 0; j < R; j++) d[j] = __builtin_nontemporal_load((const vec *)(pl + offB[j + 1]));
   };
   auto stencil = [&](const vec &c, const vec &old, const vec &cU, const vec &cD, const vec &nU, const vec &nC, const vec &nD,
                      const vec &pU, const vec &pC, const vec &pD) {
      const Real cUm = lane_from_lower<true>(cU[V - 1]), cUp = lane_from_upper<true>(cU[0]);
      const Real cDm = lane_from_lower<true>(cD[V - 1]), cDp = lane_from_upper<true>(cD[0]);
      const Real nCm = lane_from_lower<true>(nC[V - 1]), nCp = lane_from_upper<true>(nC[0]);
      const Real pCm = lane_from_lower<true>(pC[V - 1]), pCp = lane_from_upper<true>(pC[0]);
      Real a2 = a2_; // opaque per row, as in k_tb2_fcc
      asm volatile("" : "+s"(a2));
      vec o;
#pragma unroll
      for (int i = 0; i < V; i++) {
         const int im = i > 0 ? i - 1 : 0, ip = i < V - 1 ? i + 1 : V - 1;
         Real p = a1 * c[i] - old[i];
         p = p + a2 * nU[i];                               // +x+y
         p = p + a2 * pD[i];                               // -x-y
         p = p + a2 * ((i == V - 1) ? cUp : cU[ip]);       // +y+z
         p = p + a2 * ((i == 0) ? cDm : cD[im]);           // -y-z
         p = p + a2 * ((i == V - 1) ? nCp : nC[ip]);       // +x+z
         p = p + a2 * ((i == 0) ? pCm : pC[im]);           // -x-z
         p = p + a2 * nD[i];                               // +x-y
         p = p + a2 * pU[i];                               // -x+y
         p = p + a2 * ((i == 0) ? cUm : cU[im]);           // +y-z
         p = p + a2 * ((i == V - 1) ? cDp : cD[ip]);       // -y+z
         p = p + a2 * ((i == 0) ? nCm : nC[im]);           // +x-z
         p = p + a2 * ((i == V - 1) ? pCp : pC[ip]);       // -x+z
         o[i] = p;
      }
      return o;
   };
   // Four u^n plane buffers: x1-1, x1, x1+1 and the plane x1+2 still in flight -- with one barrier per plane and one
   // workgroup per CU there is nobody to hide a load issued in the same turn it is needed, so planes are requested two
   // turns ahead (x1+3 goes into the buffer of x1-1 as soon as stage 1 is done with it); likewise two u^{n-1} buffers.
   // The x loop is unrolled by four so that these roles are renamings; the three u^{n+1} planes rotate by moves.
   vec b0[R + 2], b1[R + 2], b2[R + 2], b3[R + 2], A0[R], A1[R], Bold[R];
   vec Vm[R + 2], Vc[R + 2], Vn[R + 2];
   loadB(xs - 2, b0);
   loadB(xs - 1, b1);
   loadB(xs, b2);
   loadB(min(xs + 1, xe + 1), b3);
   loadA(xs - 1, A0);
   loadA(xs, A1);
#pragma unroll
   for (int j = 0; j < R + 2; j++) { Vm[j] = vec{}; Vc[j] = vec{}; }
   auto turn = [&](int x1, vec(&Pm)[R + 2], vec(&Pc)[R + 2], vec(&Pn)[R + 2], vec(&Ac)[R]) {
      const int buf = (x1 - xs + 1) & 1;
      // stage 1: u^{n+1}(x1) on the wave's own rows yo .. yo+R-1, published to the workgroup
#pragma unroll
      for (int j = 0; j < R; j++) {
         Vn[j + 1] = stencil(Pc[j + 1], Ac[j], Pc[j + 2], Pc[j], Pn[j + 2], Pn[j + 1], Pn[j], Pm[j + 2], Pm[j + 1], Pm[j]);
         *(vec *)&sV[buf][w * R + j][lane * V] = Vn[j + 1];
      }
#pragma unroll
      for (int r = 0; r < R; r++) Bold[r] = Pm[r + 1];
      if (x1 + 3 <= xe + 1) loadB(x1 + 3, Pm);
      if (x1 + 2 <= xe) loadA(x1 + 2, Ac);
      __syncthreads();
      if (inner) {
         Vn[0] = *(const vec *)&sV[buf][w * R - 1][lane * V];
         Vn[R + 1] = *(const vec *)&sV[buf][w * R + R][lane * V];
      }
      if (x1 >= xs && x1 < xe) {
         Real *pc = (Real *)tp.C + (int64_t)x1 * plane;
#pragma unroll
         for (int r = 0; r < R; r++)
            if (core_col && core_row[r]) __builtin_nontemporal_store(Vn[r + 1], (vec *)(pc + offB[r + 1]));
      }
      // stage 2 (inner waves): u^{n+2}(x1-1) from u^{n+1} planes x1-2, x1-1, x1; its old value is u^n(x1-1)
      if (inner && x1 - 1 >= xs) {
         Real *pd = (Real *)tp.D + (int64_t)(x1 - 1) * plane;
#pragma unroll
         for (int r = 0; r < R; r++) {
            const vec o = stencil(Vc[r + 1], Bold[r], Vc[r + 2], Vc[r], Vn[r + 2], Vn[r + 1], Vn[r], Vm[r + 2], Vm[r + 1], Vm[r]);
            if (core_col && core_row[r]) __builtin_nontemporal_store(o, (vec *)(pd + offB[r + 1]));
         }
      }
#pragma unroll
      for (int j = 0; j < R + 2; j++) { Vm[j] = Vc[j]; Vc[j] = Vn[j]; }
   };
   for (int x1 = xs - 1; x1 <= xe; x1 += 4) {
      turn(x1, b0, b1, b2, A0);
      if (x1 + 1 > xe) break;
      turn(x1 + 1, b1, b2, b3, A1);
      if (x1 + 2 > xe) break;
      turn(x1 + 2, b2, b3, b0, A0);
      if (x1 + 3 > xe) break;
      turn(x1 + 3, b3, b0, b1, A1);
   }
}

// ---------------------------------------------------------------------------------------------------------------
// k_tb2_fcc_w -- k_tb2_fcc_x's scheme (waves stacked in y, u^{n+1} halo rows exchanged through LDS, first / last wave pure
// halo providers) with HALF the vector arithmetic, in both numerics, and for either storage order (round 5).
//   CPU-exact (SG = false): the reference accumulates  p += a2 * u1[neighbour]  twelve times per cell (cpu_engine.h:200-218), and
//     every cell is the neighbour of twelve others: the product a2 * u is rounded ONCE per cell here -- each u^n row becomes
//     N = a2 * u when it first serves as a neighbour row, each new u^{n+1} row likewise before it is published -- and a cell
//     update is a1 * c - old followed by twelve plain adds of N values in the reference's order: 14 vector operations instead
//     of 26, the same bits (the products are the same roundings, -ffp-contract=off).  Only the centre rows are also kept raw
//     (for a1 * c and as stage 2's old value).  k_tb2_fcc_x prevented exactly this sharing (the compiler tried it across all
//     rows at once and spilled); done by hand it costs 10 more vectors of state per lane than k_tb2_fcc_x.
//   GPU-safeguarded (SG = true): the rows stay raw; a cell update is upd13<true>: the towards-zero pairwise tree of
//     gpu_engine.h:257-267 and two FMAs, 13 operations.
//   SWZ: the grid is stored with the file's x and z axes exchanged (Engine::swz); the kernel works in storage coordinates, only
//     the ORDER in which the neighbours enter the sum (or the tree) follows the file's axes.
// 64-lane row segments only; R rows per wave, WT waves (WT - 2 of them own rows).
// ---------------------------------------------------------------------------------------------------------------
template <typename Real, int R, int WT, bool SG, bool SWZ = false>
__global__ __launch_bounds__(64 * WT) void k_tb2_fcc_w(Tb2Params tp, Real a1, Real a2) {
   typedef typename VecOf<Real>::type vec;
   constexpr int V = VecOf<Real>::V, W = 64 * V;
   __shared__ __attribute__((aligned(16))) Real sV[2][WT * R][W];
   const uint32_t t = tp.tiles ? (uint32_t)tp.tiles[blockIdx.x] : blockIdx.x;
   const int zt = t % tp.nzt, yt = (t / tp.nzt) % tp.nyt, xc = t / (tp.nzt * tp.nyt);
   const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
   const bool inner = w >= 1 && w <= WT - 2;
   const int ze0 = tp.z_begin - V + zt * (W - 2 * V);
   const int yo = tp.y_begin + (yt * (WT - 2) + (w - 1)) * R;   // first own row of this wave (wave 0: the R rows above the tile)
   const int xs = tp.x_begin + xc * tp.chunk, xe = min(xs + tp.chunk, tp.x_end);
   const int P = tp.P;
   const int64_t plane = tp.plane;
   const int zc = min(max(ze0 + lane * V, 0), P - V);
   uint32_t offB[R + 2];                                      // rows yo-1 .. yo+R
#pragma unroll
   for (int i = 0; i < R + 2; i++) offB[i] = (uint32_t)min(max(yo - 1 + i, 0), tp.Ny - 1) * (uint32_t)P + (uint32_t)zc;
   const int z_end = tp.z_end ? tp.z_end : tp.Nz - tp.z_begin, y_end = tp.y_end ? tp.y_end : tp.Ny - tp.y_begin;
   const bool core_col = inner && (lane >= 1 && lane <= 62) && (ze0 + lane * V + V - 1 < z_end);
   bool core_row[R];
#pragma unroll
   for (int r = 0; r < R; r++) core_row[r] = (yo + r < y_end);

   auto loadB = [&](int x, vec *d) {
      const Real *pl = (const Real *)tp.B + (int64_t)x * plane;
#pragma unroll
      for (int i = 0; i < R + 2; i++) d[i] = *(const vec *)(pl + offB[i]);
   };
   auto loadA = [&](int x, vec *d) { // rows yo .. yo+R-1
      const Real *pl = (const Real *)tp.A + (int64_t)x * plane;
#pragma unroll
      for (int j = 0; j < R; j++) d[j] = __builtin_nontemporal_load((const vec *)(pl + offB[j + 1]));
   };
   // c, old: the cell's own row and its value two steps back, RAW; the eight neighbour rows in N form (a2 * u, SG: raw):
   // cU / cD: rows y+1 / y-1 of plane x; n*: plane x+1, p*: plane x-1 (U, C, D = rows y+1, y, y-1)
   auto stencil = [&](const vec &c, const vec &old, const vec &cU, const vec &cD, const vec &nU, const vec &nC, const vec &nD,
                      const vec &pU, const vec &pC, const vec &pD) {
      const Real cUm = lane_from_lower<true>(cU[V - 1]), cUp = lane_from_upper<true>(cU[0]);
      const Real cDm = lane_from_lower<true>(cD[V - 1]), cDp = lane_from_upper<true>(cD[0]);
      const Real nCm = lane_from_lower<true>(nC[V - 1]), nCp = lane_from_upper<true>(nC[0]);
      const Real pCm = lane_from_lower<true>(pC[V - 1]), pCp = lane_from_upper<true>(pC[0]);
      vec o;
#pragma unroll
      for (int i = 0; i < V; i++) {
         const int im = i > 0 ? i - 1 : 0, ip = i < V - 1 ? i + 1 : V - 1;
         const Real S[12] = {nU[i], pD[i], (i == V - 1) ? cUp : cU[ip], (i == 0) ? cDm : cD[im],
                             (i == V - 1) ? nCp : nC[ip], (i == 0) ? pCm : pC[im], nD[i], pU[i],
                             (i == 0) ? cUm : cU[im], (i == V - 1) ? cDp : cD[ip], (i == 0) ? nCm : nC[im], (i == V - 1) ? pCp : pC[ip]};
         if constexpr (SG) {
            Real nb[12];
#pragma unroll
            for (int k = 0; k < 12; k++) nb[k] = S[FccOrder<SWZ>::p[k]];
            o[i] = upd13<true>(a1, a2, c[i], old[i], nb);
         } else {
            Real p = a1 * c[i] - old[i]; // cpu_engine.h:200-218 with the products a2 * u1[...] taken from the N rows
#pragma unroll
            for (int k = 0; k < 12; k++) p = p + S[FccOrder<SWZ>::p[k]];
            o[i] = p;
         }
      }
      return o;
   };
   auto to_n = [&](const vec &v) -> vec { // N form of a raw row
      if constexpr (SG) return v;
      else return v * a2;
   };
   // Four u^n plane buffers whose roles rotate with the turn (x1-1, x1, x1+1 in N form, x1+2 in flight: planes are requested two
   // turns ahead, as in k_tb2_fcc_x), each with the raw copy of its R centre rows; two u^{n-1} buffers.  The x loop is unrolled by
   // four so that these roles are renamings; the three u^{n+1} planes rotate by moves.
   vec b0[R + 2], b1[R + 2], b2[R + 2], b3[R + 2], c0[R], c1[R], c2[R], c3[R], A0[R], A1[R];
   vec Vm[R + 2], Vc[R + 2], Vn[R + 2], rVc[R], rVn[R];
   loadB(xs - 2, b0);
   loadB(xs - 1, b1);
   loadB(xs, b2);
   loadB(min(xs + 1, xe + 1), b3);
   loadA(xs - 1, A0);
   loadA(xs, A1);
#pragma unroll
   for (int r = 0; r < R; r++) { c0[r] = b0[r + 1]; c1[r] = b1[r + 1]; rVc[r] = vec{}; }
#pragma unroll
   for (int j = 0; j < R + 2; j++) { b0[j] = to_n(b0[j]); b1[j] = to_n(b1[j]); Vm[j] = vec{}; Vc[j] = vec{}; Vn[j] = vec{}; }
   // one turn: x1 = plane of the u^{n+1} values computed; Pm / Pc / Pn = u^n planes x1-1 / x1 / x1+1 (Pn arrives raw), Cm / Cc /
   // Cn their raw centre rows; Vm / Vc = u^{n+1} planes x1-2 / x1-1 in N form, Vn receives plane x1
   auto turn = [&](int x1, vec(&Pm)[R + 2], vec(&Pc)[R + 2], vec(&Pn)[R + 2], vec(&Cm)[R], vec(&Cc)[R], vec(&Cn)[R], vec(&Ac)[R]) {
      const int buf = (x1 - xs + 1) & 1;
#pragma unroll
      for (int r = 0; r < R; r++) Cn[r] = Pn[r + 1];
#pragma unroll
      for (int j = 0; j < R + 2; j++) Pn[j] = to_n(Pn[j]);
      // stage 1: u^{n+1}(x1) on the wave's own rows yo .. yo+R-1, published to the workgroup in N form
#pragma unroll
      for (int j = 0; j < R; j++) {
         rVn[j] = stencil(Cc[j], Ac[j], Pc[j + 2], Pc[j], Pn[j + 2], Pn[j + 1], Pn[j], Pm[j + 2], Pm[j + 1], Pm[j]);
         Vn[j + 1] = to_n(rVn[j]);
         *(vec *)&sV[buf][w * R + j][lane * V] = Vn[j + 1];
      }
      if (x1 + 3 <= xe + 1) loadB(x1 + 3, Pm);
      if (x1 + 2 <= xe) loadA(x1 + 2, Ac);
      __syncthreads();
      if (inner) {
         Vn[0] = *(const vec *)&sV[buf][w * R - 1][lane * V];
         Vn[R + 1] = *(const vec *)&sV[buf][w * R + R][lane * V];
      }
      if (x1 >= xs && x1 < xe) {
         Real *pc = (Real *)tp.C + (int64_t)x1 * plane;
#pragma unroll
         for (int r = 0; r < R; r++)
            if (core_col && core_row[r]) __builtin_nontemporal_store(rVn[r], (vec *)(pc + offB[r + 1]));
      }
      // stage 2 (inner waves): u^{n+2}(x1-1) from u^{n+1} planes x1-2, x1-1, x1; its old value is u^n(x1-1)
      if (inner && x1 - 1 >= xs) {
         Real *pd = (Real *)tp.D + (int64_t)(x1 - 1) * plane;
#pragma unroll
         for (int r = 0; r < R; r++) {
            const vec o = stencil(rVc[r], Cm[r], Vc[r + 2], Vc[r], Vn[r + 2], Vn[r + 1], Vn[r], Vm[r + 2], Vm[r + 1], Vm[r]);
            if (core_col && core_row[r]) __builtin_nontemporal_store(o, (vec *)(pd + offB[r + 1]));
         }
      }
#pragma unroll
      for (int j = 0; j < R + 2; j++) { Vm[j] = Vc[j]; Vc[j] = Vn[j]; }
#pragma unroll
      for (int r = 0; r < R; r++) rVc[r] = rVn[r];
   };
   for (int x1 = xs - 1; x1 <= xe; x1 += 4) {
      turn(x1, b0, b1, b2, c0, c1, c2, A0);
      if (x1 + 1 > xe) break;
      turn(x1 + 1, b1, b2, b3, c1, c2, c3, A1);
      if (x1 + 2 > xe) break;
      turn(x1 + 2, b2, b3, b0, c2, c3, c0, A0);
      if (x1 + 3 > xe) break;
      turn(x1 + 3, b3, b0, b1, c3, c0, c1, A1);
   }
}

// host-side launcher, defined (and the kernel instantiated) in pf_tb2_fcc.hip, which is built with its own flags
template <typename Real> void launch_tb2_fcc(hipStream_t s, const Tb2Params &tp, Real a1, Real a2, int lw, uint32_t nblocks, bool sg = false, bool swz = false);

// ---------------------------------------------------------------------------------------------------------------
// k_tb1_tile -- ONE 7-point air update of the tiles k_tb2_reg must leave alone (a boundary node, a source or the ABC
// shell within one cell of their core), with the same tile geometry, out of place: A = u^{n-1}, B = u^n -> C = u^{n+1}.
// Cells whose skip-mask bit is set (boundary nodes) are not written: the boundary pass writes them afterwards.
// Inside the box of tiles there are no ghost cells and no ABC cells, so none of that is handled here.
// ---------------------------------------------------------------------------------------------------------------
// HL: halo lanes per side of a row segment (1; 2 for the fp64 tiles of k_tb3, whose three stages need three halo cells)
template <typename Real, int R, int WY, int LW = 64, bool SG = false, bool SWZ = false, int HL = 1>
__global__ __launch_bounds__(64 * WY) void k_tb1_tile(Tb2Params tp, Real a1, Real a2) {
   typedef typename VecOf<Real>::type vec;
   static_assert(LW == 64 || LW == 32 || LW == 16, "row segments are 64, 32 or 16 lanes wide");
   constexpr int V = VecOf<Real>::V, W = LW * V, NSUB = 64 / LW;
   // a handful of dirty tiles (the source's) is a launch of a few workgroups marching 16 planes each, all latency: xsub
   // workgroups per tile shorten the march (each starts with its own two-plane prologue)
   const uint32_t nsplit = tp.xsub > 1 ? (uint32_t)tp.xsub : 1u, tb = blockIdx.x / nsplit, piece = blockIdx.x % nsplit;
   const uint32_t t = tp.tiles ? (uint32_t)tp.tiles[tb] : tb;
   const int zt = t % tp.nzt, yt = (t / tp.nzt) % tp.nyt, xc = t / (tp.nzt * tp.nyt);
   const int wlane = threadIdx.x & 63, w = threadIdx.x >> 6;
   const int lane = wlane % LW, sub = wlane / LW;
   const int ze0 = tp.z_begin - HL * V + zt * (W - 2 * HL * V);
   const int yo = tp.y_begin + ((yt * WY + w) * NSUB + sub) * R;
   const int xs0 = tp.x_begin + xc * tp.chunk, xe0 = min(xs0 + tp.chunk, tp.x_end);
   const int plen = (xe0 - xs0 + (int)nsplit - 1) / (int)nsplit;
   const int xs = xs0 + (int)piece * plen, xe = min(xs + plen, xe0);
   if (xs >= xe) return;
   const int P = tp.P;
   const int64_t plane = tp.plane;
   const int zc = min(max(ze0 + lane * V, 0), P - V);
   int64_t off[R + 2];                                        // rows yo-1 .. yo+R
#pragma unroll
   for (int i = 0; i < R + 2; i++) off[i] = (int64_t)min(max(yo - 1 + i, 0), tp.Ny - 1) * P + zc;
   const int z_end = tp.z_end ? tp.z_end : tp.Nz - tp.z_begin, y_end = tp.y_end ? tp.y_end : tp.Ny - tp.y_begin;
   const bool core_col = (lane >= HL && lane <= LW - 1 - HL) && (ze0 + lane * V + V - 1 < z_end);
   bool core_row[R];
#pragma unroll
   for (int r = 0; r < R; r++) core_row[r] = (yo + r < y_end);
   auto loadB = [&](int x, vec *d) {
      const Real *pl = (const Real *)tp.B + (int64_t)x * plane;
#pragma unroll
      for (int i = 0; i < R + 2; i++) d[i] = *(const vec *)(pl + off[i]);
   };
   vec Bp[R], Bc[R + 2], Bn[R + 2];
   {
      vec t2[R + 2];
      loadB(xs - 1, t2);
#pragma unroll
      for (int r = 0; r < R; r++) Bp[r] = t2[r + 1];
      loadB(xs, Bc);
   }
   for (int x = xs; x < xe; x++) {
      loadB(x + 1, Bn);
      const Real *pa = (const Real *)tp.A + (int64_t)x * plane;
      Real *pc = (Real *)tp.C + (int64_t)x * plane;
      const uint8_t *pm = tp.mask + (((int64_t)x * plane) >> 3);
#pragma unroll
      for (int r = 0; r < R; r++) {
         const vec old = __builtin_nontemporal_load((const vec *)(pa + off[r + 1]));
         const uint32_t bits = (uint32_t)pm[off[r + 1] >> 3] >> (uint32_t)(off[r + 1] & 7);
         const vec c = Bc[r + 1];
         const Real lf = lane_from_lower<true>(c[V - 1]);
         const Real rt = lane_from_upper<true>(c[0]);
         vec o;
#pragma unroll
         for (int i = 0; i < V; i++) {
            const Real zp = (i == V - 1) ? rt : c[i < V - 1 ? i + 1 : V - 1];
            const Real zm = (i == 0) ? lf : c[i > 0 ? i - 1 : 0];
            if constexpr (SWZ) o[i] = upd7<SG>(a1, a2, c[i], old[i], zp, zm, Bc[r + 2][i], Bc[r][i], Bn[r + 1][i], Bp[r][i]);
            else o[i] = upd7<SG>(a1, a2, c[i], old[i], Bn[r + 1][i], Bp[r][i], Bc[r + 2][i], Bc[r][i], zp, zm);
         }
         if (core_col && core_row[r]) {
            if ((bits & ((1u << V) - 1u)) == 0u) __builtin_nontemporal_store(o, (vec *)(pc + off[r + 1]));
            else {
#pragma unroll
               for (int i = 0; i < V; i++)
                  if (!((bits >> i) & 1u)) pc[off[r + 1] + i] = o[i];
            }
         }
      }
#pragma unroll
      for (int r = 0; r < R; r++) Bp[r] = Bc[r + 1];
#pragma unroll
      for (int i = 0; i < R + 2; i++) Bc[i] = Bn[i];
   }
}

// ---------------------------------------------------------------------------------------------------------------
// k_air_zstrip -- single-step 7-point air update (virtual ghost shell + ABC loss, as k_air_cart_lean) of the thin
// column strips z in [0, zl) and [zr, P) left over next to the temporally blocked box: one thread per 16-byte vector.
// Out of place: u^{n-1} from u0s, u^{n+1} to u0.  Same expression order as the marching kernels (bit-identical).
// ---------------------------------------------------------------------------------------------------------------
template <typename Real> struct ZStripParams {
   const Real *u1, *u0s;
   Real *u0;
   const uint8_t *mask;
   int64_t plane;
   int32_t Nx, Ny, Nz, P;
   int32_t x_begin, x_end;   // planes [x_begin, x_end)
   int32_t zl, zr;           // strips [0, zl) and [zr, P), multiples of 4
   int32_t first, last;
   // boundary nodes inside the strips (optional; zvec == null: the boundary-list kernel does them, masked cells keep
   // their old value here).  The strips' nodes are numbered in strip order (x, y, z) and their adjacency bits / lossy-list
   // positions are stored in that order; zvec[(x*Ny + y)*nv + v] = (number of the vector's first node << 4) | one bit per
   // cell of the vector that holds a node: 4 bytes per 16-byte vector instead of a 128-byte line of the skip mask plus a
   // 128-byte line of cell -> node indices per row
   const uint32_t *zvec;
   const uint16_t *adjv;     // [strip node]
   const int32_t *lossy;     // [strip node] position in the lossy arrays or -1
   Real *u0b;                // lossy nodes: the RIGID result is left here (and in the grid); the branch ODEs follow in extra threads of the k_boundary launch,
                             // dense over the compact lossy arrays
   Real sl2;
};

template <typename Real, bool SG = false, bool SWZ = false>
__global__ __launch_bounds__(256) void k_air_zstrip(ZStripParams<Real> zp, Real a1, Real a2, Real l, int xchunk) {
   // thread = one 16-byte vector of one row; it marches xchunk planes with the x neighbours in registers, so every
   // 128-byte line of u1 / u0s next to the strip is fetched once (a thread-per-cell version re-fetched the x neighbours
   // from other XCDs: 7x the compulsory bytes)
   typedef typename VecOf<Real>::type vec;
   constexpr int V = VecOf<Real>::V;
   const int nl = zp.zl / V, nr = (zp.P - zp.zr) / V, nv = nl + nr;
   const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t >= (int64_t)(zp.Ny - 2) * nv) return;
   const int v = (int)(t % nv);
   const int y = 1 + (int)(t / nv);
   const int xs = zp.x_begin + blockIdx.y * xchunk, xe = min(xs + xchunk, zp.x_end);
   const int z0 = v < nl ? v * V : zp.zr + (v - nl) * V;
   const int Nx = zp.Nx, Ny = zp.Ny, Nz = zp.Nz, P = zp.P;
   auto rowsrc = [&](int yy) { return yy == 0 ? 2 : (yy == Ny - 1 ? Ny - 3 : yy); };
   auto planesrc = [&](int xx) { return (zp.first && xx == 0) ? 2 : ((zp.last && xx == Nx - 1) ? Nx - 3 : xx); };
   const int64_t off = (int64_t)y * P + z0, offp = (int64_t)rowsrc(y + 1) * P + z0, offm = (int64_t)rowsrc(y - 1) * P + z0;
   const int zzN = Nz - 1 - z0; // position of the ghost column Nz-1 relative to this vector
   const bool yq = (y == 1 || y == Ny - 2);
   uint32_t ghost = 0; // ghost / pad columns of this vector
#pragma unroll
   for (int i = 0; i < V; i++) ghost |= (z0 + i == 0 || z0 + i >= Nz - 1) ? (1u << i) : 0u;
   auto centre = [&](int x, vec &c, Real &lf, Real &rt) { // a row of plane x with its z neighbours, ghost columns patched
      const Real *pc = zp.u1 + (int64_t)planesrc(x) * zp.plane;
      c = *(const vec *)(pc + off);
      lf = z0 > 0 ? pc[off - 1] : Real(0);
      rt = z0 + V < P ? pc[off + V] : Real(0);
      if (V == 4) { // same patches as load_own_row of k_air_cart_lean
         if (z0 == 0) c[0] = c[2];
         if (zzN == 1) c[1] = lf;
         if (zzN == 2) c[2] = c[0];
         if (zzN == 3) c[3] = c[1];
      } else {
         if (z0 == 0) c[0] = rt;
         if (zzN == 1) c[1] = lf;
      }
      if (zzN == V) rt = c[V - 2];
   };
   vec cm, c, cp;
   Real lf, rt, lfn, rtn, dl, dr;
   centre(xs - 1, cm, dl, dr);
   centre(xs, c, lf, rt);
   for (int x = xs; x < xe; x++) {
      centre(x + 1, cp, lfn, rtn);
      const Real *pc = zp.u1 + (int64_t)x * zp.plane;
      const vec yp = *(const vec *)(pc + offp), ym = *(const vec *)(pc + offm);
      const vec old = *(const vec *)(zp.u0s + (int64_t)x * zp.plane + off);
      const uint32_t zrec = zp.zvec ? zp.zvec[((int64_t)x * Ny + y) * nv + v] : 0u;
      const uint32_t bits = zp.zvec ? (ghost | (zrec & 15u)) : ((zp.mask[((int64_t)x * zp.plane + off) >> 3] >> (off & 7)) & ((1u << V) - 1u));
      const int qxy = (((zp.first && x == 1) || (zp.last && x == Nx - 2)) ? 1 : 0) + (yq ? 1 : 0);
      vec o;
#pragma unroll
      for (int i = 0; i < V; i++) {
         const Real zpv = (i == V - 1) ? rt : c[i < V - 1 ? i + 1 : V - 1];
         const Real zmv = (i == 0) ? lf : c[i > 0 ? i - 1 : 0];
         Real p = SWZ ? upd7<SG>(a1, a2, c[i], old[i], zpv, zmv, yp[i], ym[i], cp[i], cm[i]) // (file order of the neighbours: axes exchanged in storage)
                      : upd7<SG>(a1, a2, c[i], old[i], cp[i], cm[i], yp[i], ym[i], zpv, zmv);
         const int Q = qxy + ((z0 + i == 1 || z0 + i == Nz - 2) ? 1 : 0);
         if (Q > 0) p = abc_loss<SG>(p, old[i], l * (Real)Q); // (cpu_engine.h:225-229 incl. the double literal of :228; SG: gpu_engine.h:351-365)
         if ((bits >> i) & 1u) {
            p = old[i]; // ghost / pad column, or a boundary node that the list kernel updates
            const int32_t nb = ((zrec >> i) & 1u) ? (int32_t)((zrec >> 4) + __popc(zrec & ((1u << i) - 1u))) : -1;
            if (nb >= 0) { // boundary node (its number in strip order): rigid update from the registers (k_boundary's expression), then the FD branches
               const uint32_t adj = zp.adjv[nb];
               const Real nbk[6] = {SWZ ? zpv : cp[i], SWZ ? zmv : cm[i], yp[i], ym[i], SWZ ? cp[i] : zpv, SWZ ? cm[i] : zmv}; // adjacency-bit (file) order
               p = upd_rigid<SG, 6>(a2, zp.sl2, adj, c[i], old[i], nbk);
               const int32_t li = zp.lossy[nb];
               if (li >= 0) zp.u0b[li] = p;
            }
         }
         o[i] = p;
      }
      *(vec *)(zp.u0 + (int64_t)x * zp.plane + off) = o;
      cm = c; c = cp; lf = lfn; rt = rtn;
   }
}

// ---------------------------------------------------------------------------------------------------------------
// k_zstrip_fcc -- 13-point counterpart of k_air_zstrip for the folded FCC grid: single-step update (+ ABC loss) of the thin
// column strips z in [0, zl) and [zr, P) beside the temporally blocked box, out of place (u^{n-1} from u0s, u^{n+1} to
// u0).  The ghost shell of u1 is in memory here (the 13-point path keeps the flip kernels), so rows y-1 .. y+1 and the
// columns next to a vector are simply loaded.  Boundary nodes keep their old value (the list kernel writes them).
// Same accumulation order as k_air_fcc (bit-identical).
// ---------------------------------------------------------------------------------------------------------------
template <typename Real, bool SG = false, bool SWZ = false>
__global__ __launch_bounds__(256) void k_zstrip_fcc(ZStripParams<Real> zp, Real a1, Real a2, Real l, int xchunk, int fold) {
   typedef typename VecOf<Real>::type vec;
   constexpr int V = VecOf<Real>::V;
   const int nl = zp.zl / V, nr = (zp.P - zp.zr) / V, nv = nl + nr;
   const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t >= (int64_t)(zp.Ny - 2) * nv) return;
   const int v = (int)(t % nv);
   const int y = 1 + (int)(t / nv);
   const int xs = zp.x_begin + blockIdx.y * xchunk, xe = min(xs + xchunk, zp.x_end);
   const int z0 = v < nl ? v * V : zp.zr + (v - nl) * V;
   const int Nx = zp.Nx, Ny = zp.Ny, Nz = zp.Nz, P = zp.P;
   const int64_t off = (int64_t)y * P + z0;
   struct Row { vec c; Real lf, rt; };
   auto load_row = [&](const Real *pl, int64_t o) {
      Row r;
      r.c = *(const vec *)(pl + o);
      r.lf = z0 > 0 ? pl[o - 1] : Real(0);
      r.rt = z0 + V < P ? pl[o + V] : Real(0);
      return r;
   };
   auto load_plane = [&](int x, Row *d) { // rows y-1, y, y+1
      const Real *pl = zp.u1 + (int64_t)x * zp.plane;
      d[0] = load_row(pl, off - P); d[1] = load_row(pl, off); d[2] = load_row(pl, off + P);
   };
   auto lo = [&](const Row &r, int i) { return i == 0 ? r.lf : r.c[i > 0 ? i - 1 : 0]; };
   auto hi = [&](const Row &r, int i) { return i == V - 1 ? r.rt : r.c[i < V - 1 ? i + 1 : V - 1]; };
   const int qy = (y == 1 || (!fold && y == Ny - 2)) ? 1 : 0;
   Row pm[3], pc[3], pn[3];
   load_plane(xs - 1, pm);
   load_plane(xs, pc);
   for (int x = xs; x < xe; x++) {
      load_plane(x + 1, pn);
      const vec old = *(const vec *)(zp.u0s + (int64_t)x * zp.plane + off);
      const uint32_t bits = (zp.mask[((int64_t)x * zp.plane + off) >> 3] >> (off & 7)) & ((1u << V) - 1u);
      const int qxy = (((zp.first && x == 1) || (zp.last && x == Nx - 2)) ? 1 : 0) + qy;
      vec o;
#pragma unroll
      for (int i = 0; i < V; i++) {
         // storage order of the twelve neighbours; they enter the sum in the FILE's order (FccOrder: axes exchanged in storage)
         const Real S[12] = {pn[2].c[i], pm[0].c[i], hi(pc[2], i), lo(pc[0], i), hi(pn[1], i), lo(pm[1], i),
                             pn[0].c[i], pm[2].c[i], lo(pc[2], i), hi(pc[0], i), lo(pn[1], i), hi(pm[1], i)};
         Real nb[12];
#pragma unroll
         for (int k = 0; k < 12; k++) nb[k] = S[FccOrder<SWZ>::p[k]];
         Real p = upd13<SG>(a1, a2, pc[1].c[i], old[i], nb);
         const int Q = qxy + ((z0 + i == 1 || z0 + i == Nz - 2) ? 1 : 0);
         if (Q > 0) p = abc_loss<SG>(p, old[i], l * (Real)Q); // ABC loss (cpu_engine.h:225-229 incl. the double literal of :228; SG: gpu_engine.h:351-365)
         if ((bits >> i) & 1u) p = old[i]; // ghost / pad column or a boundary node
         o[i] = p;
      }
      *(vec *)(zp.u0 + (int64_t)x * zp.plane + off) = o;
#pragma unroll
      for (int r = 0; r < 3; r++) { pm[r] = pc[r]; pc[r] = pn[r]; }
   }
}

} // namespace pf
