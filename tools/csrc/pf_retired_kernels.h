// pf_retired_kernels.h -- interior kernels that were built, measured against the production kernels and NOT kept on the
// product path ("things tried"): the generic fused kernel with in-kernel rigid update by mask-bit ranking
// (k_air_fused: correct, issue-bound, 2.4x slower), the LDS-DMA landing-ring kernel (k_air_cart_lds: same plateau as the lean
// kernel), the 13-point lean kernel (k_air_fcc_lean: slower than k_air_fcc, register pressure) and the one-thread-per-cell
// kernel (k_air_naive).  They lived in libpffdtd_hip.so behind air_variant 9-14 / 20-24 / 33 / 35 until round 3 and are kept
// here, outside the product library, for reference; nothing launches them.
#pragma once
#include "pf_kernels.h"
#include "pf_air_fused.h"

#pragma clang fp contract(off)

namespace pf {

template <bool FMA, typename Real> __device__ __forceinline__ Real acc(Real p, Real a2, Real x) { // (the retired PF_NUM_FMA mode)
   if (FMA) return __builtin_fma(a2, x, p);
   return p + a2 * x;
}
// LeanParams as the retired lean-family kernels knew it
struct LeanParamsOld : LeanParams { const uint8_t *adj; int32_t debug; };

struct FusedParams {
   const void *u1;
   void *u0;
   const uint8_t *mask;      // padded layout, boundary nodes only (bit jj&7 of byte jj>>3)
   const uint16_t *adj;      // adjacency words of the sorted boundary-node list
   const int32_t *segstart;  // [(ix*Ny+iy)*nzt + seg] -> index of the first boundary node of that row segment
   int64_t plane;            // Ny*P
   int32_t Nx, Ny, Nz, P;
   int32_t x_begin, x_end, chunk;
   int32_t nzt, nyt, nxc, swizzle;
   int32_t first, last;      // slab holds the global ix=0 / ix=Nx-1 ghost plane
   int32_t fold, parity;     // fcc_flag==2 / fcc_flag==1 (1 + parity of the global ix of plane 0)
   int32_t do_abc, do_rigid;
};

template <typename Real, bool FCC, int R, int WY, bool FMA>
__global__ __launch_bounds__(64 * WY) void k_air_fused(FusedParams fp, Real a1, Real a2, Real sl2, Real l) {
   typedef typename VecOf<Real>::type vec;
   constexpr int V = VecOf<Real>::V;
   constexpr int W = 64 * V;
   constexpr int LROW = W + 4;           // W values, L edge, R edge, 2 pad (keeps rows 16-byte aligned)
   constexpr int NROWS = 2 * WY + 2;     // halo_top, {first,last} of every wave, halo_bot
   constexpr int NW = FCC ? R + 2 : R;   // rows kept for the previous plane
   static_assert(WY >= 2, "the top and the bottom wave each carry one workgroup halo row");
   __shared__ __attribute__((aligned(16))) Real lds[2][NROWS][LROW];

   const Real *__restrict__ u1 = (const Real *)fp.u1;
   Real *__restrict__ u0 = (Real *)fp.u0;
   const uint32_t total = (uint32_t)fp.nzt * fp.nyt * fp.nxc;
   uint32_t b = blockIdx.x;
   if (fp.swizzle == 2) { if (!xcd_band(blockIdx.x, (uint32_t)fp.nzt * fp.nyt, (uint32_t)fp.nxc, b)) return; }
   else if (fp.swizzle) b = xcd_swizzle(b, total);
   const int zt = b % fp.nzt;
   const int yt = (b / fp.nzt) % fp.nyt;
   const int xc = b / (fp.nzt * fp.nyt);
   const int lane = threadIdx.x & 63;
   const int w = threadIdx.x >> 6;
   const int Nx = fp.Nx, Ny = fp.Ny, Nz = fp.Nz, P = fp.P;
   const int64_t plane = fp.plane;
   const int z0 = (zt * 64 + lane) * V;
   const bool active = z0 < P;
   const int zl = active ? z0 : 0;
   const int yb0 = 1 + yt * WY * R;
   const int y0 = yb0 + w * R;
   const int xs = fp.x_begin + xc * fp.chunk;
   const int xe = min(xs + fp.chunk, fp.x_end);

   // ---- virtual ghost shell: source row / plane of a load ----
   auto rowsrc = [&](int y) {
      y = min(y, Ny - 1);
      if (y == 0) return 2;
      if (y == Ny - 1) return fp.fold ? Ny - 2 : Ny - 3;
      return y;
   };
   auto planesrc = [&](int x) {
      if (fp.first && x == 0) return 2;
      if (fp.last && x == Nx - 1) return Nx - 3;
      return x;
   };
   const bool need_l = (lane == 0) && (z0 > 0);
   const bool need_r = (lane == 63) && (z0 + V < P);
   const bool r_is_ghost = (z0 + V == Nz - 1);       // right neighbour column is the ghost column: value = column Nz-3
   const bool has_z0 = active && (z0 == 0);
   const bool ownsN = active && (z0 <= Nz - 1) && (Nz - 1 < z0 + V);
   const int zzN = Nz - 1 - z0;

   // one row of a plane: 16 B per lane + wave-edge columns, ghost columns replaced by their mirror cells
   auto load_row = [&](const Real *pl, uint32_t off, vec &v, Real &L, Real &Rr) {
      v = *(const vec *)(pl + off);
      L = need_l ? pl[off - 1] : Real(0);
      Real rr = need_r ? pl[off + V] : Real(0);
      const Real gN = ownsN ? pl[off - (uint32_t)zl + (uint32_t)(Nz - 3)] : Real(0);
      if (V == 4) {
         if (has_z0) v[0] = v[2];
      } else {
         const Real g0 = has_z0 ? pl[off + 2] : Real(0);
         if (has_z0) v[0] = g0;
      }
      if (ownsN) {
#pragma unroll
         for (int i = 0; i < V; i++)
            if (i == zzN) v[i] = gN;
      }
      if (need_r && r_is_ghost) rr = v[V - 2];
      Rr = rr;
   };

   // in-plane offsets (elements) of the rows this lane touches
   uint32_t ro[R], so[R];
   bool valid[R];
#pragma unroll
   for (int r = 0; r < R; r++) {
      ro[r] = (uint32_t)rowsrc(y0 + r) * (uint32_t)P + (uint32_t)zl;
      so[r] = (uint32_t)min(y0 + r, Ny - 1) * (uint32_t)P + (uint32_t)zl;
      valid[r] = active && (y0 + r <= Ny - 2);
   }
   const uint32_t ro_above = (uint32_t)rowsrc(y0 - 1) * (uint32_t)P + (uint32_t)zl;
   const uint32_t ro_below = (uint32_t)rowsrc(y0 + R) * (uint32_t)P + (uint32_t)zl;
   const bool top_wave = (w == 0), bot_wave = (w == WY - 1);

   // per-lane cell classes that do not depend on x: z ghost / pad columns, z shell, y shell
   bool skipz[V];
   int qz[V];
#pragma unroll
   for (int i = 0; i < V; i++) {
      const int z = z0 + i;
      skipz[i] = (z == 0) || (z >= Nz - 1);
      qz[i] = (z == 1 || z == Nz - 2) ? 1 : 0;
   }
   int qy[R];
#pragma unroll
   for (int r = 0; r < R; r++) {
      const int y = y0 + r;
      qy[r] = (y == 1 || (!fp.fold && y == Ny - 2)) ? 1 : 0;
   }

   // ---- plane windows ----
   vec prev[NW], cur[R + 2], nxt[R + 2], nn[R];
   Real prevL[NW], prevR[NW], curL[R + 2], curR[R + 2], nxtL[R + 2], nxtR[R + 2], nnL[R], nnR[R];
   vec hv = {};            // workgroup halo row of the plane in flight (top / bottom wave only)
   Real hL = 0, hR = 0;
   vec old[R], oldn[R];
   uint32_t mb[R], mbn[R];

   auto load_own = [&](int x, vec *dst, Real *dL, Real *dR) { // dst[r] <- own rows of plane x
      const Real *pl = u1 + (int64_t)planesrc(x) * plane;
#pragma unroll
      for (int r = 0; r < R; r++) load_row(pl, ro[r], dst[r], dL[r], dR[r]);
   };
   auto load_halo = [&](int x) { // workgroup halo row of plane x into hv/hL/hR (top and bottom wave)
      const Real *pl = u1 + (int64_t)planesrc(x) * plane;
      if (top_wave) load_row(pl, ro_above, hv, hL, hR);
      if (bot_wave) load_row(pl, ro_below, hv, hL, hR);
   };
   auto load_old = [&](int x, vec *d, uint32_t *m) {
      const Real *po = u0 + (int64_t)x * plane;
      const uint8_t *pm = fp.mask + (((int64_t)x * plane) >> 3);
#pragma unroll
      for (int r = 0; r < R; r++) {
         d[r] = *(const vec *)(po + so[r]);
         m[r] = pm[so[r] >> 3];
      }
   };
   {  // prologue: windows of planes xs-1 and xs straight from global memory, own rows of xs+1 in flight
      const Real *pm = u1 + (int64_t)planesrc(xs - 1) * plane;
      const Real *pc = u1 + (int64_t)planesrc(xs) * plane;
      if (FCC) {
         load_row(pm, ro_above, prev[0], prevL[0], prevR[0]);
         load_row(pm, ro_below, prev[NW - 1], prevL[NW - 1], prevR[NW - 1]);
      }
#pragma unroll
      for (int r = 0; r < R; r++) load_row(pm, ro[r], prev[FCC ? r + 1 : r], prevL[FCC ? r + 1 : r], prevR[FCC ? r + 1 : r]);
      load_row(pc, ro_above, cur[0], curL[0], curR[0]);
      load_row(pc, ro_below, cur[R + 1], curL[R + 1], curR[R + 1]);
#pragma unroll
      for (int r = 0; r < R; r++) load_row(pc, ro[r], cur[r + 1], curL[r + 1], curR[r + 1]);
      load_own(xs + 1, &nxt[1], &nxtL[1], &nxtR[1]);
      load_halo(xs + 1);
      load_old(xs, old, mb);
   }

   for (int x = xs; x < xe; x++) {
      const bool more = (x + 1 < xe);
      // (A) publish the edge rows of plane x+1 (loaded one iteration ago) for the neighbouring waves
      Real(*S)[LROW] = lds[(x + 1) & 1];
      {
         *(vec *)&S[1 + 2 * w][lane * V] = nxt[1];
         *(vec *)&S[2 + 2 * w][lane * V] = nxt[R];
         if (lane == 0) { S[1 + 2 * w][W] = nxtL[1]; S[2 + 2 * w][W] = nxtL[R]; }
         if (lane == 63) { S[1 + 2 * w][W + 1] = nxtR[1]; S[2 + 2 * w][W + 1] = nxtR[R]; }
         if (top_wave || bot_wave) {
            const int hr = top_wave ? 0 : NROWS - 1;
            *(vec *)&S[hr][lane * V] = hv;
            if (lane == 0) S[hr][W] = hL;
            if (lane == 63) S[hr][W + 1] = hR;
         }
      }
      // (B) next loads: own rows (+ workgroup halo row) of plane x+2, old state and mask of plane x+1
      if (more) {
         load_own(x + 2, nn, nnL, nnR);
         load_halo(x + 2);
         load_old(x + 1, oldn, mbn);
      }
      __syncthreads();
      // (C) rows above / below my strip in plane x+1
      nxt[0] = *(const vec *)&S[2 * w][lane * V];
      nxtL[0] = S[2 * w][W];
      nxtR[0] = S[2 * w][W + 1];
      nxt[R + 1] = *(const vec *)&S[2 * w + 3][lane * V];
      nxtL[R + 1] = S[2 * w + 3][W];
      nxtR[R + 1] = S[2 * w + 3][W + 1];

      // (D) update plane x
      Real *po = u0 + (int64_t)x * plane;
      const int qx = ((fp.first && x == 1) || (fp.last && x == Nx - 2)) ? 1 : 0;
      auto zlo = [&](const vec &v, Real edge) { // value at z-1 for every element
         Real zm = lane_from_lower<true>(v[V - 1]);
         if (lane == 0) zm = edge;
         vec s;
#pragma unroll
         for (int i = 0; i < V; i++) s[i] = (i == 0) ? zm : v[i > 0 ? i - 1 : 0];
         return s;
      };
      auto zhi = [&](const vec &v, Real edge) { // value at z+1 for every element
         Real zp = lane_from_upper<true>(v[0]);
         if (lane == 63) zp = edge;
         vec s;
#pragma unroll
         for (int i = 0; i < V; i++) s[i] = (i == V - 1) ? zp : v[i < V - 1 ? i + 1 : V - 1];
         return s;
      };
#pragma unroll
      for (int r = 0; r < R; r++) {
         const int j = r + 1;
         const int y = y0 + r;
         const vec c = cur[j];
         // neighbour values in the reference's accumulation order
         constexpr int NNB = FCC ? 12 : 6;
         vec nb[NNB];
         if (!FCC) {
            nb[0] = nxt[j];                 // +NzNy
            nb[1] = prev[r];                // -NzNy
            nb[2] = cur[j + 1];             // +Nz
            nb[3] = cur[j - 1];             // -Nz
            nb[4] = zhi(c, curR[j]);        // +1
            nb[5] = zlo(c, curL[j]);        // -1
         } else {
            const int jp = r + 1;           // own row inside prev[] (R+2 rows)
            nb[0] = nxt[j + 1];                          // +NzNy+Nz
            nb[1] = prev[jp - 1];                        // -NzNy-Nz
            nb[2] = zhi(cur[j + 1], curR[j + 1]);        // +Nz+1
            nb[3] = zlo(cur[j - 1], curL[j - 1]);        // -Nz-1
            nb[4] = zhi(nxt[j], nxtR[j]);                // +NzNy+1
            nb[5] = zlo(prev[jp], prevL[jp]);            // -NzNy-1
            nb[6] = nxt[j - 1];                          // +NzNy-Nz
            nb[7] = prev[jp + 1];                        // -NzNy+Nz
            nb[8] = zlo(cur[j + 1], curL[j + 1]);        // +Nz-1
            nb[9] = zhi(cur[j - 1], curR[j - 1]);        // -Nz+1
            nb[10] = zlo(nxt[j], nxtL[j]);               // +NzNy-1
            nb[11] = zhi(prev[jp], prevR[jp]);           // -NzNy+1
         }
         const uint32_t bits = valid[r] ? ((mb[r] >> (so[r] & 7u)) & ((1u << V) - 1u)) : 0u;
         const bool row_has_bn = __ballot(bits != 0) != 0ull;
         vec o;
         if (!(fp.do_rigid && row_has_bn)) {
            // air cells only (boundary cells keep their value for the separate rigid kernel)
#pragma unroll
            for (int i = 0; i < V; i++) {
               Real p = a1 * c[i] - old[r][i];
#pragma unroll
               for (int k = 0; k < NNB; k++) p = acc<FMA>(p, a2, nb[k][i]);
               o[i] = p;
            }
         } else {
            // row segment with boundary nodes: per-cell centre coefficient and neighbour weights.
            // rank of a boundary cell = #mask bits before it in this row segment (wave prefix count)
            uint32_t before = 0;
#pragma unroll
            for (int i = 0; i < V; i++) {
               const unsigned long long m = __ballot((bits >> i) & 1u);
               before += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            }
            const int32_t seg0 = fp.segstart[((int64_t)x * Ny + y) * fp.nzt + zt];
            uint32_t inl = 0;
#pragma unroll
            for (int i = 0; i < V; i++) {
               const bool isbn = (bits >> i) & 1u;
               uint32_t adjw = (1u << NNB) - 1u;
               if (isbn) adjw = fp.adj[seg0 + (int32_t)(before + inl)];
               inl += isbn ? 1u : 0u;
               const Real two = 2.0;
               const Real cc = isbn ? (two - sl2 * (Real)__popc(adjw)) : a1; // b1 (cpu_engine.h:245) | a1
               Real p = cc * c[i] - old[r][i];
#pragma unroll
               for (int k = 0; k < NNB; k++) {
                  const Real wk = ((adjw >> k) & 1u) ? a2 : Real(0); // == a2*(Real)bit exactly
                  p = FMA ? __builtin_fma(wk, nb[k][i], p) : p + wk * nb[k][i];
               }
               o[i] = p;
            }
         }
         // ABC loss on the outermost interior shell; u2ba is the old value of this very cell
         if (fp.do_abc) {
            bool any = false;
#pragma unroll
            for (int i = 0; i < V; i++) any = any || ((qx + qy[r] + qz[i]) > 0);
            if (__ballot(any) != 0ull) {
#pragma unroll
               for (int i = 0; i < V; i++) {
                  const int Q = qx + qy[r] + qz[i];
                  if (Q > 0 && !((bits >> i) & 1u)) {
                     const Real lQ = l * (Real)Q;
                     const Real num = o[i] + lQ * old[r][i];
                     o[i] = (Real)((double)num / (1.0 + (double)lQ)); // double literal of cpu_engine.h:228
                  }
               }
            }
         }
#pragma unroll
         for (int i = 0; i < V; i++) {
            bool keep = skipz[i];
            if (fp.parity) keep = keep || (((x + y + z0 + i + (fp.parity - 1)) & 1) != 0);
            if (!fp.do_rigid) keep = keep || ((bits >> i) & 1u);
            if (keep) o[i] = old[r][i];
         }
         if (valid[r]) *(vec *)(po + so[r]) = o;
      }
      // (E) rotate the windows
      if (FCC) {
#pragma unroll
         for (int j = 0; j < R + 2; j++) { prev[j] = cur[j]; prevL[j] = curL[j]; prevR[j] = curR[j]; }
      } else {
#pragma unroll
         for (int r = 0; r < R; r++) { prev[r] = cur[r + 1]; }
      }
#pragma unroll
      for (int j = 0; j < R + 2; j++) { cur[j] = nxt[j]; curL[j] = nxtL[j]; curR[j] = nxtR[j]; }
#pragma unroll
      for (int r = 0; r < R; r++) {
         nxt[r + 1] = nn[r]; nxtL[r + 1] = nnL[r]; nxtR[r + 1] = nnR[r];
         old[r] = oldn[r]; mb[r] = mbn[r];
      }
   }
}

// first boundary node of every (row, z-segment): lower bound of the segment's first padded index in the sorted list
__global__ void k_segstart(const int64_t *__restrict__ bn, int64_t Nb, int32_t *__restrict__ segstart, int64_t nrows,
                           int32_t nzt, int64_t P, int32_t segw) {
   const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (t >= nrows * nzt) return;
   const int64_t row = t / nzt;
   const int32_t seg = (int32_t)(t % nzt);
   const int64_t key = row * P + (int64_t)seg * segw;
   int64_t lo = 0, hi = Nb;
   while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (bn[mid] < key) lo = mid + 1; else hi = mid;
   }
   segstart[t] = (int32_t)lo;
}

// one adjacency byte per padded cell for the fused rigid update: 0x80 | adjacency bits at boundary nodes, 0 elsewhere
__global__ void k_adj_dense_set(uint8_t *__restrict__ dense, const int64_t *__restrict__ idx, const uint16_t *__restrict__ adj, int64_t n) {
   const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (i < n) dense[idx[i]] = (uint8_t)(0x80u | (adj[i] & 0x3fu));
}

// boundary-node-only mask in the padded layout
__global__ void k_mask_zero(uint8_t *__restrict__ mask, int64_t nbytes) {
   const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   if (i < nbytes) mask[i] = 0;
}


// =============================================================================================================
// k_air_cart_lds -- 7-point kernel with the u1 stream landing in LDS by DMA (global_load_lds_dwordx4).
// Same fused work as k_air_cart_lean (virtual ghost shell + air + ABC), different data path: the tile of plane
// x+2 ((R*WY + 2) rows x 1 KiB) is in flight into a 3-slot LDS ring while plane x is computed from LDS, so the
// bytes in flight cost no VGPRs and the register footprint (hence occupancy) is that of the bare stencil.
// Row layout in LDS: [W values | V-wide chunk left of the segment | V-wide chunk right of it]; the two chunks give
// the z neighbours of the first / last lane and are fetched by a 2-lane DMA.
// Per plane x: [vmcnt(0); barrier] -> plane x+1 has landed for every wave; [DMA plane x+2 into slot (x+2)%3 =
// the slot plane x-1 just left] [prefetch old/mask of x+1 into registers] [read c/up/down/left/right of plane x and
// the centre of plane x+1 from LDS] [update, ABC, store]; plane x-1's centre values stay in registers.
// =============================================================================================================
#define PF_GPTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define PF_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

template <typename Real, int R, int WY, bool FMA>
__global__ __launch_bounds__(64 * WY) void k_air_cart_lds(LeanParams fp, Real a1, Real a2, Real l) {
   typedef typename VecOf<Real>::type vec;
   constexpr int V = VecOf<Real>::V;
   constexpr int W = 64 * V;
   constexpr int TY = R * WY;
   constexpr int ROW = W + 2 * V;
   static_assert(WY >= 2, "top and bottom wave each carry one workgroup halo row");
   static_assert(R == 1, "EXPERIMENTAL: with more than one DMA row per wave hipcc (ROCm 7.2) clobbers the exec-mask SGPR pair "
                         "of the 2-lane chunk DMA with the M0 staging register (NaNs on MI355X); R=1 is validated bit-exact");
   __shared__ __attribute__((aligned(16))) Real ring[3][TY + 2][ROW];

   const Real *__restrict__ u1 = (const Real *)fp.u1;
   Real *__restrict__ u0 = (Real *)fp.u0;
   const uint32_t total = (uint32_t)fp.nzt * fp.nyt * fp.nxc;
   uint32_t b = blockIdx.x;
   if (fp.swizzle == 2) { if (!xcd_band(blockIdx.x, (uint32_t)fp.nzt * fp.nyt, (uint32_t)fp.nxc, b)) return; }
   else if (fp.swizzle) b = xcd_swizzle(b, total);
   const int zt = b % fp.nzt;
   const int yt = (b / fp.nzt) % fp.nyt;
   const int xc = b / (fp.nzt * fp.nyt);
   const int lane = threadIdx.x & 63;
   const int w = threadIdx.x >> 6;
   const int Nx = fp.Nx, Ny = fp.Ny, Nz = fp.Nz, P = fp.P;
   const int64_t plane = fp.plane;
   const int zseg = zt * W;
   const int z0 = zseg + lane * V;
   const bool active = z0 < P;
   const int zl = active ? z0 : 0;
   const int y0 = 1 + (yt * WY + w) * R;
   const int xs = fp.x_begin + xc * fp.chunk;
   const int xe = min(xs + fp.chunk, fp.x_end);
   const bool top_wave = (w == 0), bot_wave = (w == WY - 1);
   const bool halo_wave = top_wave || bot_wave;
   const int halo_lrow = top_wave ? 0 : TY + 1;

   auto rowsrc = [&](int y) {
      y = min(y, Ny - 1);
      if (y == 0) return 2;
      if (y == Ny - 1) return Ny - 3;
      return y;
   };
   auto planesrc = [&](int x) {
      if (fp.first && x == 0) return 2;
      if (fp.last && x == Nx - 1) return Nx - 3;
      return x;
   };
   uint32_t rb[R], so[R];
   bool valid[R];
#pragma unroll
   for (int r = 0; r < R; r++) {
      rb[r] = (uint32_t)rowsrc(y0 + r) * (uint32_t)P;
      so[r] = (uint32_t)min(y0 + r, Ny - 1) * (uint32_t)P + (uint32_t)zl;
      valid[r] = active && (y0 + r <= Ny - 2);
   }
   const uint32_t rb_halo = (uint32_t)rowsrc(top_wave ? y0 - 1 : y0 + R) * (uint32_t)P;
   const int chunk_col = (lane == 0) ? zseg - V : zseg + W;   // 2-lane chunk DMA (lane 0: left, lane 1: right)
   // LDS columns of the z neighbours of this lane's first / last element
   const int col_left = (lane == 0) ? W + V - 1 : lane * V - 1;
   const int col_right = (lane == 63) ? W + V : lane * V + V;

   const int zzN = Nz - 1 - z0;
   const bool fix0 = (z0 == 0);
   const bool fixR = (zzN == V);
   uint32_t qzbits = 0;
#pragma unroll
   for (int i = 0; i < V; i++)
      if (active && (z0 + i == 1 || z0 + i == Nz - 2)) qzbits |= 1u << i;
   const bool wave_has_qz = __ballot(qzbits != 0) != 0ull;

   auto dma_plane = [&](int x) {
      const Real *pl = u1 + (int64_t)planesrc(x) * plane;
      Real(*S)[ROW] = ring[x % 3];
#pragma unroll
      for (int r = 0; r < R; r++) {
         __builtin_amdgcn_global_load_lds(PF_GPTR(pl + rb[r] + zl), PF_LPTR(&S[1 + w * R + r][0]), 16, 0, 0);
         if (lane < 2)
            __builtin_amdgcn_global_load_lds(PF_GPTR(pl + (int64_t)rb[r] + chunk_col), PF_LPTR(&S[1 + w * R + r][W]), 16, 0, 0);
      }
      if (halo_wave) __builtin_amdgcn_global_load_lds(PF_GPTR(pl + rb_halo + zl), PF_LPTR(&S[halo_lrow][0]), 16, 0, 0);
   };
   auto load_old = [&](int x, vec *d, uint32_t *m) {
      const Real *po = u0 + (int64_t)x * plane;
      const uint8_t *pm = fp.mask + (((int64_t)x * plane) >> 3);
#pragma unroll
      for (int r = 0; r < R; r++) {
         d[r] = *(const vec *)(po + so[r]);
         m[r] = pm[so[r] >> 3];
      }
   };

   vec prev[R], old[R], oldn[R];
   uint32_t mb[R], mbn[R];
   {  // prologue: plane xs-1 centre rows to registers, planes xs and xs+1 to the ring
      const Real *pm = u1 + (int64_t)planesrc(xs - 1) * plane;
#pragma unroll
      for (int r = 0; r < R; r++) prev[r] = *(const vec *)(pm + rb[r] + zl);
      dma_plane(xs);
      dma_plane(xs + 1);
      load_old(xs, old, mb);
   }

   for (int x = xs; x < xe; x++) {
      const bool more = (x + 1 < xe);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (more) {
         dma_plane(x + 2);
         load_old(x + 1, oldn, mbn);
      }
      Real(*SC)[ROW] = ring[x % 3];
      Real(*SN)[ROW] = ring[(x + 1) % 3];
      Real *po = u0 + (int64_t)x * plane;
      const bool qx = (fp.first && x == 1) || (fp.last && x == Nx - 2);
#pragma unroll
      for (int r = 0; r < R; r++) {
         const int lr = 1 + w * R + r;
         vec c = *(const vec *)&SC[lr][lane * V];
         const vec ym = *(const vec *)&SC[lr - 1][lane * V];
         const vec yp = *(const vec *)&SC[lr + 1][lane * V];
         const vec xp = *(const vec *)&SN[lr][lane * V];
         Real lf = SC[lr][col_left];
         Real rt = SC[lr][col_right];
         const vec craw = c;
         // virtual z ghost columns (column 0 mirrors column 2, column Nz-1 mirrors Nz-3)
         if (V == 4) {
            if (fix0) c[0] = c[2];
            if (zzN == 1) c[1] = lf;
            if (zzN == 2) c[2] = c[0];
            if (zzN == 3) c[3] = c[1];
         } else {
            if (fix0) c[0] = rt;
            if (zzN == 1) c[1] = lf;
         }
         if (fixR) rt = c[V - 2];
         const uint32_t bits = mb[r] >> (so[r] & 7u);
         vec o;
#pragma unroll
         for (int i = 0; i < V; i++) {
            const Real zp = (i == V - 1) ? rt : c[i < V - 1 ? i + 1 : V - 1];
            const Real zm = (i == 0) ? lf : c[i > 0 ? i - 1 : 0];
            Real p = a1 * c[i] - old[r][i];
            p = acc<FMA>(p, a2, xp[i]);       // +NzNy
            p = acc<FMA>(p, a2, prev[r][i]);  // -NzNy
            p = acc<FMA>(p, a2, yp[i]);       // +Nz
            p = acc<FMA>(p, a2, ym[i]);       // -Nz
            p = acc<FMA>(p, a2, zp);          // +1
            p = acc<FMA>(p, a2, zm);          // -1
            o[i] = p;
         }
         if (fp.do_abc) {
            const int y = y0 + r;
            const int qxy = (qx ? 1 : 0) + ((y == 1 || y == Ny - 2) ? 1 : 0);
            if (qxy > 0 || wave_has_qz) {
#pragma unroll
               for (int i = 0; i < V; i++) {
                  const bool zq = (qzbits >> i) & 1u;
                  if (qxy > 0 || __ballot(zq) != 0ull) {
                     const int Q = qxy + (zq ? 1 : 0);
                     if (Q > 0) {
                        const Real lQ = l * (Real)Q;
                        const Real num = o[i] + lQ * old[r][i];
                        o[i] = (Real)((double)num / (1.0 + (double)lQ)); // double literal of cpu_engine.h:228
                     }
                  }
               }
            }
         }
#pragma unroll
         for (int i = 0; i < V; i++)
            if ((bits >> i) & 1u) o[i] = old[r][i];
         if (valid[r]) *(vec *)(po + so[r]) = o;
         prev[r] = craw;
      }
#pragma unroll
      for (int r = 0; r < R; r++) { old[r] = oldn[r]; mb[r] = mbn[r]; }
   }
}


// =============================================================================================================
// k_air_fcc_lean -- 13-point FCC counterpart of k_air_cart_lean (folded grid fcc_flag 2 and, through the parity bits
// of the skip-mask, the checkerboard grid fcc_flag 1): virtual ghost shell + air update + ABC loss in one pass.
// Neighbour / accumulation order of cpu_engine.h:204-216.  Every row that is used with a z offset carries its two
// z-neighbour columns (lf, rt); rows shared between waves (and the workgroup halo rows) travel through a 4-slot LDS
// ring as [W values | lf of lane 0 | rt of lane 63]: plane x+1 is published at the top of iteration x, one barrier,
// then the rows above / below this wave's strip are read for planes x-1, x and x+1.
// =============================================================================================================
template <typename Real, int R, int WY, bool FMA>
__global__ __launch_bounds__(64 * WY) void k_air_fcc_lean(LeanParams fp, Real a1, Real a2, Real l, int fold) {
   typedef typename VecOf<Real>::type vec;
   constexpr int V = VecOf<Real>::V;
   constexpr int W = 64 * V;
   constexpr int LROW = W + 4;
   constexpr int NROWS = 2 * WY + 2;
   static_assert(WY >= 2, "top and bottom wave each carry one workgroup halo row");
   __shared__ __attribute__((aligned(16))) Real lds[4][NROWS][LROW];

   const Real *__restrict__ u1 = (const Real *)fp.u1;
   Real *__restrict__ u0 = (Real *)fp.u0;
   const uint32_t total = (uint32_t)fp.nzt * fp.nyt * fp.nxc;
   uint32_t b = blockIdx.x;
   if (fp.swizzle == 2) { if (!xcd_band(blockIdx.x, (uint32_t)fp.nzt * fp.nyt, (uint32_t)fp.nxc, b)) return; }
   else if (fp.swizzle) b = xcd_swizzle(b, total);
   const int zt = b % fp.nzt;
   const int yt = (b / fp.nzt) % fp.nyt;
   const int xc = b / (fp.nzt * fp.nyt);
   const int lane = threadIdx.x & 63;
   const int w = threadIdx.x >> 6;
   const int Nx = fp.Nx, Ny = fp.Ny, Nz = fp.Nz, P = fp.P;
   const int64_t plane = fp.plane;
   const int z0 = (zt * 64 + lane) * V;
   const bool active = z0 < P;
   const int zl = active ? z0 : 0;
   const int y0 = 1 + (yt * WY + w) * R;
   const int xs = fp.x_begin + xc * fp.chunk;
   const int xe = min(xs + fp.chunk, fp.x_end);
   const bool top_wave = (w == 0), bot_wave = (w == WY - 1);
   const bool halo_wave = top_wave || bot_wave;
   const int halo_slot = top_wave ? 0 : NROWS - 1;

   auto rowsrc = [&](int y) {
      y = min(y, Ny - 1);
      if (y == 0) return 2;
      if (y == Ny - 1) return fold ? Ny - 2 : Ny - 3;
      return y;
   };
   auto planesrc = [&](int x) {
      if (fp.first && x == 0) return 2;
      if (fp.last && x == Nx - 1) return Nx - 3;
      return x;
   };
   uint32_t ro[R], so[R];
   bool valid[R];
#pragma unroll
   for (int r = 0; r < R; r++) {
      ro[r] = (uint32_t)rowsrc(y0 + r) * (uint32_t)P + (uint32_t)zl;
      so[r] = (uint32_t)min(y0 + r, Ny - 1) * (uint32_t)P + (uint32_t)zl;
      valid[r] = active && (y0 + r <= Ny - 2);
   }
   const uint32_t ro_halo = (uint32_t)rowsrc(top_wave ? y0 - 1 : y0 + R) * (uint32_t)P + (uint32_t)zl;
   const int zzN = Nz - 1 - z0;
   const bool fix0 = (z0 == 0);
   const bool fixR = (zzN == V);
   uint32_t qzbits = 0;
#pragma unroll
   for (int i = 0; i < V; i++)
      if (active && (z0 + i == 1 || z0 + i == Nz - 2)) qzbits |= 1u << i;
   const bool wave_has_qz = __ballot(qzbits != 0) != 0ull;

   struct Row { vec v; Real lf, rt; };
   auto load_row = [&](const Real *pl, uint32_t off) {
      Row q;
      q.v = *(const vec *)(pl + off);
      q.lf = pl[off - 1];
      q.rt = pl[off + V];
      if (V == 4) {
         if (fix0) q.v[0] = q.v[2];
         if (zzN == 1) q.v[1] = q.lf;
         if (zzN == 2) q.v[2] = q.v[0];
         if (zzN == 3) q.v[3] = q.v[1];
      } else {
         if (fix0) q.v[0] = q.rt;
         if (zzN == 1) q.v[1] = q.lf;
      }
      if (fixR) q.rt = q.v[V - 2];
      return q;
   };
   auto zlo = [&](const Row &q) { // value at z-1 of every element
      vec s;
#pragma unroll
      for (int i = 0; i < V; i++) s[i] = (i == 0) ? q.lf : q.v[i > 0 ? i - 1 : 0];
      return s;
   };
   auto zhi = [&](const Row &q) { // value at z+1 of every element
      vec s;
#pragma unroll
      for (int i = 0; i < V; i++) s[i] = (i == V - 1) ? q.rt : q.v[i < V - 1 ? i + 1 : V - 1];
      return s;
   };
   auto publish = [&](int x, const Row *rows, const Row &h) {
      Real(*S)[LROW] = lds[x & 3];
      *(vec *)&S[1 + 2 * w][lane * V] = rows[0].v;
      *(vec *)&S[2 + 2 * w][lane * V] = rows[R - 1].v;
      if (lane == 0) { S[1 + 2 * w][W] = rows[0].lf; S[2 + 2 * w][W] = rows[R - 1].lf; }
      if (lane == 63) { S[1 + 2 * w][W + 1] = rows[0].rt; S[2 + 2 * w][W + 1] = rows[R - 1].rt; }
      if (halo_wave) {
         *(vec *)&S[halo_slot][lane * V] = h.v;
         if (lane == 0) S[halo_slot][W] = h.lf;
         if (lane == 63) S[halo_slot][W + 1] = h.rt;
      }
   };
   auto read_halo = [&](int x, int srow, bool with_lr) { // row `srow` of plane x from the ring
      Real(*S)[LROW] = lds[x & 3];
      Row q;
      q.v = *(const vec *)&S[srow][lane * V];
      q.lf = Real(0);
      q.rt = Real(0);
      if (with_lr) {
         // z neighbours of the first / last element: own lanes' data, or the published wave-edge columns
         const Real l0 = S[srow][W], r63 = S[srow][W + 1];
         const Real lm = S[srow][lane == 0 ? 0 : lane * V - 1];
         const Real rp = S[srow][lane == 63 ? W - 1 : lane * V + V];
         q.lf = (lane == 0) ? l0 : lm;
         q.rt = (lane == 63) ? r63 : rp;
         if (fixR) q.rt = q.v[V - 2]; // my right neighbour is the ghost column: mirror of column Nz-3
      }
      return q;
   };

   Row prev[R], cur[R], nxt[R], nn[R];
   Row hv = {}, hvn = {};
   vec old[R], oldn[R];
   uint32_t mb[R], mbn[R];
   auto load_plane_own = [&](int x, Row *d) {
      const Real *pl = u1 + (int64_t)planesrc(x) * plane;
#pragma unroll
      for (int r = 0; r < R; r++) d[r] = load_row(pl, ro[r]);
   };
   auto load_halo = [&](int x) { return load_row(u1 + (int64_t)planesrc(x) * plane, ro_halo); };
   auto load_old = [&](int x, vec *d, uint32_t *m) {
      const Real *po = u0 + (int64_t)x * plane;
      const uint8_t *pm = fp.mask + (((int64_t)x * plane) >> 3);
#pragma unroll
      for (int r = 0; r < R; r++) {
         d[r] = *(const vec *)(po + so[r]);
         m[r] = pm[so[r] >> 3];
      }
   };
   {  // prologue: planes xs-1 and xs published, plane xs+1 loaded (published at the top of the first iteration)
      load_plane_own(xs - 1, prev);
      Row h = {};
      if (halo_wave) h = load_halo(xs - 1);
      publish(xs - 1, prev, h);
      load_plane_own(xs, cur);
      if (halo_wave) h = load_halo(xs);
      publish(xs, cur, h);
      load_plane_own(xs + 1, nxt);
      if (halo_wave) hv = load_halo(xs + 1);
      load_old(xs, old, mb);
   }

   for (int x = xs; x < xe; x++) {
      const bool more = (x + 1 < xe);
      publish(x + 1, nxt, hv);
      if (more) {
         load_plane_own(x + 2, nn);
         if (halo_wave) hvn = load_halo(x + 2);
         load_old(x + 1, oldn, mbn);
      }
      __syncthreads();
      const Row pa = read_halo(x - 1, 2 * w, false), pb = read_halo(x - 1, 2 * w + 3, false);
      const Row ca = read_halo(x, 2 * w, true), cb = read_halo(x, 2 * w + 3, true);
      const Row na = read_halo(x + 1, 2 * w, false), nb_ = read_halo(x + 1, 2 * w + 3, false);

      Real *po = u0 + (int64_t)x * plane;
      const bool qx = (fp.first && x == 1) || (fp.last && x == Nx - 2);
#pragma unroll
      for (int r = 0; r < R; r++) {
         const Row &c = cur[r];
         const Row &cu = (r == R - 1) ? cb : cur[r < R - 1 ? r + 1 : R - 1]; // row y+1 of plane x
         const Row &cd = (r == 0) ? ca : cur[r > 0 ? r - 1 : 0];             // row y-1
         const vec nu = (r == R - 1) ? nb_.v : nxt[r < R - 1 ? r + 1 : R - 1].v;
         const vec nd = (r == 0) ? na.v : nxt[r > 0 ? r - 1 : 0].v;
         const vec pu = (r == R - 1) ? pb.v : prev[r < R - 1 ? r + 1 : R - 1].v;
         const vec pd = (r == 0) ? pa.v : prev[r > 0 ? r - 1 : 0].v;
         vec nbv[12];
         nbv[0] = nu;            // +NzNy+Nz
         nbv[1] = pd;            // -NzNy-Nz
         nbv[2] = zhi(cu);       // +Nz+1
         nbv[3] = zlo(cd);       // -Nz-1
         nbv[4] = zhi(nxt[r]);   // +NzNy+1
         nbv[5] = zlo(prev[r]);  // -NzNy-1
         nbv[6] = nd;            // +NzNy-Nz
         nbv[7] = pu;            // -NzNy+Nz
         nbv[8] = zlo(cu);       // +Nz-1
         nbv[9] = zhi(cd);       // -Nz+1
         nbv[10] = zlo(nxt[r]);  // +NzNy-1
         nbv[11] = zhi(prev[r]); // -NzNy+1
         const uint32_t bits = mb[r] >> (so[r] & 7u);
         vec o;
#pragma unroll
         for (int i = 0; i < V; i++) {
            Real p = a1 * c.v[i] - old[r][i];
#pragma unroll
            for (int k = 0; k < 12; k++) p = acc<FMA>(p, a2, nbv[k][i]);
            o[i] = p;
         }
         if (fp.do_abc) {
            const int y = y0 + r;
            const int qxy = (qx ? 1 : 0) + ((y == 1 || (!fold && y == Ny - 2)) ? 1 : 0);
            if (qxy > 0 || wave_has_qz) {
#pragma unroll
               for (int i = 0; i < V; i++) {
                  const bool zq = (qzbits >> i) & 1u;
                  if (qxy > 0 || __ballot(zq) != 0ull) {
                     const int Q = qxy + (zq ? 1 : 0);
                     if (Q > 0) {
                        const Real lQ = l * (Real)Q;
                        const Real num = o[i] + lQ * old[r][i];
                        o[i] = (Real)((double)num / (1.0 + (double)lQ)); // double literal of cpu_engine.h:228
                     }
                  }
               }
            }
         }
#pragma unroll
         for (int i = 0; i < V; i++)
            if ((bits >> i) & 1u) o[i] = old[r][i];
         if (valid[r]) *(vec *)(po + so[r]) = o;
      }
#pragma unroll
      for (int r = 0; r < R; r++) {
         prev[r] = cur[r];
         cur[r] = nxt[r];
         nxt[r] = nn[r];
         old[r] = oldn[r];
         mb[r] = mbn[r];
      }
      hv = hvn;
   }
}


// Naive one-thread-per-cell air kernels: debugging reference variant (air_variant 9), same arithmetic.
template <typename Real, bool FCC, bool FMA>
static __global__ void k_air_naive(const Real *__restrict__ u1, Real *__restrict__ u0, const uint8_t *__restrict__ mask,
                            Real a1, Real a2, int64_t Ny, int64_t Nz, int64_t P, int64_t plane, int32_t x_begin,
                            int32_t x_end) {
   const int64_t iz = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
   const int64_t iy = 1 + blockIdx.y;
   const int64_t ix = x_begin + blockIdx.z;
   if (iz < 1 || iz > Nz - 2 || iy > Ny - 2 || ix >= x_end) return;
   const int64_t jj = ix * plane + iy * P + iz;
   if ((mask[jj >> 3] >> (jj & 7)) & 1) return;
   Real p = a1 * u1[jj] - u0[jj];
   if (!FCC) {
      p = acc<FMA>(p, a2, u1[jj + plane]);
      p = acc<FMA>(p, a2, u1[jj - plane]);
      p = acc<FMA>(p, a2, u1[jj + P]);
      p = acc<FMA>(p, a2, u1[jj - P]);
      p = acc<FMA>(p, a2, u1[jj + 1]);
      p = acc<FMA>(p, a2, u1[jj - 1]);
   } else {
      p = acc<FMA>(p, a2, u1[jj + plane + P]);
      p = acc<FMA>(p, a2, u1[jj - plane - P]);
      p = acc<FMA>(p, a2, u1[jj + P + 1]);
      p = acc<FMA>(p, a2, u1[jj - P - 1]);
      p = acc<FMA>(p, a2, u1[jj + plane + 1]);
      p = acc<FMA>(p, a2, u1[jj - plane - 1]);
      p = acc<FMA>(p, a2, u1[jj + plane - P]);
      p = acc<FMA>(p, a2, u1[jj - plane + P]);
      p = acc<FMA>(p, a2, u1[jj + P - 1]);
      p = acc<FMA>(p, a2, u1[jj - P + 1]);
      p = acc<FMA>(p, a2, u1[jj + plane - 1]);
      p = acc<FMA>(p, a2, u1[jj - plane + 1]);
   }
   u0[jj] = p;
}

} // namespace pf
