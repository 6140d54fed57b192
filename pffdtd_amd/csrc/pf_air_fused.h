// pf_air_fused.h -- k_air_cart_lean, the fused 7-point single-step kernel: one launch per step does
//   * the ghost-shell "halo flips" (cpu_engine.h:135-172) VIRTUALLY: ghost planes / rows / columns are never
//     written; every load that would touch one reads the cell two steps inward instead (plane 0 -> 2, row 0 -> 2,
//     column Nz-1 -> Nz-3), which is exactly what the flips would have stored;
//   * the 7-point air update (cpu_engine.h:175-194), same association;
//   * the ABC loss on the outermost interior shell (cpu_engine.h:131-134,225-229): the "previous state" u2ba is
//     the old u0 the stencil has in registers anyway, Q follows from the coordinates (fdtd_data.h:636-647).
// 2.5D tiling: a workgroup = WY waves stacked along y, each lane owns R rows x V consecutive z (16 bytes) and
// marches along x with three plane windows in registers.  Own rows stream from HBM exactly once per plane
// (coalesced 16 B/lane); the rows shared between neighbouring waves go through LDS (first/last row of every
// wave + the two workgroup halo rows, double buffered, one barrier per plane).  Loads for plane x+2 are in flight
// while plane x is computed.
//
// The host enables this kernel only when its preconditions hold (see Engine::fused_ok): no boundary node in
// the ABC shell, grid >= 5 cells per axis, DPP self-test passed.  Otherwise the step falls back to the separate
// flip / air / ABC / rigid kernels of pf_kernels.h (same arithmetic, reference kernel order).
// (The generic fused kernel with in-kernel rigid update, the LDS-DMA variant and the 13-point lean kernel were measured,
// lost, and moved out of the product library in round 3: tools/csrc/pf_retired_kernels.h.)
#pragma once
#include "pf_kernels.h"

#pragma clang fp contract(off)

namespace pf {

// =============================================================================================================
// k_air_cart_lean -- the production 7-point kernel: virtual ghost shell + air update + ABC loss in one pass, with
// the instruction count per voxel kept close to the bare stencil (the retired generic k_air_fused spent most of
// its issue slots on predication).  Differences from it:
//   * z neighbours come from two extra unit-stride loads per own row (element z0-1 and z0+V of every lane; L1/L2
//     hits on the lines the 16-byte load just touched) instead of DPP shifts + lane-predicated edge loads;
//   * ghost columns are patched with per-lane constant selects, halo rows need no patching at all (they are only
//     used at the same z, and a ghost column's own cell is never updated);
//   * the skip-mask of pf_kernels.h (boundary nodes + ghost z + pad) decides which cells keep their old value; the
//     rigid boundary update stays in its own (list) kernel;
//   * halo rows of the current plane are read from LDS when needed instead of living in registers.
// Pipeline per plane x: [issue loads: own rows of x+2, old/mask of x+1] [read halo rows of x from LDS slot x&1]
// [publish first/last own row of x+1 into slot (x+1)&1] [update + store plane x] [barrier] [rotate registers].
// =============================================================================================================
struct LeanParams {
   const void *u1;
   void *u0;
   const uint8_t *mask;     // skip-mask (pf_kernels.h), padded layout
   int64_t plane;
   int32_t Nx, Ny, Nz, P;
   int32_t x_begin, x_end, chunk;
   int32_t nzt, nyt, nxc, swizzle;
   int32_t first, last;
   int32_t do_abc;
   int32_t swz;             // storage has the file's x and z axes exchanged (AirParams::swz)
   const void *u0_src;      // out-of-place step: u^{n-1} is read from here, u^{n+1} written to u0 (null: in place)
   int32_t yt0;             // first y tile of this launch (row-strip launches); nyt counts from there
   int32_t yt_split, yt_hi0; // two row strips in one launch: tiles [yt0, yt0+yt_split) and [yt_hi0, ...) (yt_split < 0: off)
   int32_t x2_begin, x2_nlo; // k_air_cart_lean, two x slabs in one launch (x2_nlo > 0): chunks [0, x2_nlo) march [x_begin, x_lo_end),
   int32_t x_lo_end;         //   the others [x2_begin, x_end)
   int32_t skip_y0, skip_y1; // rows [skip_y0, skip_y1) are computed but NOT stored (row-strip launches beside a box whose u^{n-1} is not
                             // in memory: the third step of a triple, pf_tb3.h); 0, 0: every row is stored
};

template <typename Real, int R, int WY, bool SG, bool NT = false>
__global__ __launch_bounds__(64 * WY) void k_air_cart_lean(LeanParams fp, Real a1, Real a2, Real l) {
   typedef typename VecOf<Real>::type vec;
   constexpr int V = VecOf<Real>::V;
   constexpr int W = 64 * V;
   constexpr int NROWS = 2 * WY + 2;
   static_assert(WY >= 2, "top and bottom wave each carry one workgroup halo row");
   __shared__ __attribute__((aligned(16))) Real lds[2][NROWS][W];

   const Real *__restrict__ u1 = (const Real *)fp.u1;
   Real *u0 = (Real *)fp.u0;
   const Real *u0s = fp.u0_src ? (const Real *)fp.u0_src : (const Real *)fp.u0;
   const uint32_t total = (uint32_t)fp.nzt * fp.nyt * fp.nxc;
   uint32_t b = blockIdx.x;
   if (fp.swizzle == 2) { if (!xcd_band(blockIdx.x, (uint32_t)fp.nzt * fp.nyt, (uint32_t)fp.nxc, b)) return; }
   else if (fp.swizzle) b = xcd_swizzle(b, total);
   const int zt = b % fp.nzt;
   int yt = (b / fp.nzt) % fp.nyt;
   yt = (fp.yt_split >= 0 && yt >= fp.yt_split) ? fp.yt_hi0 + (yt - fp.yt_split) : fp.yt0 + yt;
   const int xc = b / (fp.nzt * fp.nyt);
   const int lane = threadIdx.x & 63;
   const int w = threadIdx.x >> 6;
   const int Nx = fp.Nx, Ny = fp.Ny, Nz = fp.Nz, P = fp.P;
   const int64_t plane = fp.plane;
   const int z0 = (zt * 64 + lane) * V;
   const bool active = z0 < P;
   const int zl = active ? z0 : 0;
   const int y0 = 1 + (yt * WY + w) * R;
   const bool lo_slab = fp.x2_nlo > 0 && xc < fp.x2_nlo;   // two x slabs in one launch
   const int xs = (fp.x2_nlo > 0 && !lo_slab) ? fp.x2_begin + (xc - fp.x2_nlo) * fp.chunk : fp.x_begin + xc * fp.chunk;
   const int xe = min(xs + fp.chunk, lo_slab ? fp.x_lo_end : fp.x_end);
   const bool top_wave = (w == 0), bot_wave = (w == WY - 1);

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
   uint32_t ro[R], so[R];
   bool valid[R];
#pragma unroll
   for (int r = 0; r < R; r++) {
      ro[r] = (uint32_t)rowsrc(y0 + r) * (uint32_t)P + (uint32_t)zl;
      so[r] = (uint32_t)min(y0 + r, Ny - 1) * (uint32_t)P + (uint32_t)zl;
      valid[r] = active && (y0 + r <= Ny - 2) && !(y0 + r >= fp.skip_y0 && y0 + r < fp.skip_y1);
   }
   const uint32_t ro_halo = (uint32_t)rowsrc(top_wave ? y0 - 1 : y0 + R) * (uint32_t)P + (uint32_t)zl;
   const bool halo_wave = top_wave || bot_wave;
   const int halo_slot = top_wave ? 0 : NROWS - 1;

   // per-lane constants of the virtual z ghost columns (column 0 mirrors column 2, column Nz-1 mirrors Nz-3)
   const int zzN = Nz - 1 - z0;                 // position of the ghost column Nz-1 inside this lane's vector
   const bool fix0 = (z0 == 0);
   const bool fixR = (zzN == V);                // my right neighbour IS the ghost column
   // shell columns z==1 / z==Nz-2 (ABC)
   uint32_t qzbits = 0;
#pragma unroll
   for (int i = 0; i < V; i++)
      if (active && (z0 + i == 1 || z0 + i == Nz - 2)) qzbits |= 1u << i;
   const bool wave_has_qz = __ballot(qzbits != 0) != 0ull;

   // an own row with its z neighbours; ghost columns patched
   auto load_own_row = [&](const Real *pl, uint32_t off, vec &v, Real &lf, Real &rt) {
      v = *(const vec *)(pl + off);
      lf = pl[off - 1];
      rt = pl[off + V];
      if (V == 4) {
         if (fix0) v[0] = v[2];
         if (zzN == 1) v[1] = lf;
         if (zzN == 2) v[2] = v[0];
         if (zzN == 3) v[3] = v[1];
      } else {
         if (fix0) v[0] = rt;
         if (zzN == 1) v[1] = lf;
      }
      if (fixR) rt = v[V - 2];
   };

   vec prev[R], cur[R], nxt[R], nn[R], old[R], oldn[R];
   Real curL[R], curR[R], nxtL[R], nxtR[R], nnL[R], nnR[R];
   uint32_t mb[R], mbn[R];
   vec hv = {}, hvn = {};

   auto load_plane_own = [&](int x, vec *d, Real *dl, Real *dr) {
      const Real *pl = u1 + (int64_t)planesrc(x) * plane;
#pragma unroll
      for (int r = 0; r < R; r++) load_own_row(pl, ro[r], d[r], dl[r], dr[r]);
   };
   auto load_old = [&](int x, vec *d, uint32_t *m) {
      const Real *po = u0s + (int64_t)x * plane;
      const uint8_t *pm = fp.mask + (((int64_t)x * plane) >> 3);
#pragma unroll
      for (int r = 0; r < R; r++) {
         d[r] = NT ? __builtin_nontemporal_load((const vec *)(po + so[r])) : *(const vec *)(po + so[r]);
         m[r] = pm[so[r] >> 3];
      }
   };
   auto publish = [&](int x, const vec *rows, const vec &h) { // first/last own row (+ workgroup halo row) of plane x
      Real(*S)[W] = lds[x & 1];
      *(vec *)&S[1 + 2 * w][lane * V] = rows[0];
      *(vec *)&S[2 + 2 * w][lane * V] = rows[R - 1];
      if (halo_wave) *(vec *)&S[halo_slot][lane * V] = h;
   };

   {  // prologue
      const Real *pc = u1 + (int64_t)planesrc(xs) * plane;
      vec hc = {};
      if (halo_wave) hc = *(const vec *)(pc + ro_halo);
      load_plane_own(xs, cur, curL, curR);
      publish(xs, cur, hc);
      load_plane_own(xs + 1, nxt, nxtL, nxtR);
      if (halo_wave) hv = *(const vec *)(u1 + (int64_t)planesrc(xs + 1) * plane + ro_halo);
      Real dl[R], dr[R];
      load_plane_own(xs - 1, prev, dl, dr);
      load_old(xs, old, mb);
      __syncthreads();
   }

   for (int x = xs; x < xe; x++) {
      const bool more = (x + 1 < xe);
      if (more) {
         load_plane_own(x + 2, nn, nnL, nnR);
         if (halo_wave) hvn = *(const vec *)(u1 + (int64_t)planesrc(x + 2) * plane + ro_halo);
         load_old(x + 1, oldn, mbn);
      }
      Real(*S)[W] = lds[x & 1];
      const vec above = *(const vec *)&S[2 * w][lane * V];
      const vec below = *(const vec *)&S[2 * w + 3][lane * V];
      publish(x + 1, nxt, hv);

      Real *po = u0 + (int64_t)x * plane;
      const bool qx = (fp.first && x == 1) || (fp.last && x == Nx - 2);
#pragma unroll
      for (int r = 0; r < R; r++) {
         const vec c = cur[r];
         const vec ym = (r == 0) ? above : cur[r > 0 ? r - 1 : 0];
         const vec yp = (r == R - 1) ? below : cur[r < R - 1 ? r + 1 : R - 1];
         const uint32_t bits = (mb[r] >> (so[r] & 7u)) & ((1u << V) - 1u);
         vec o, zpv, zmv;
#pragma unroll
         for (int i = 0; i < V; i++) {
            zpv[i] = (i == V - 1) ? curR[r] : c[i < V - 1 ? i + 1 : V - 1];
            zmv[i] = (i == 0) ? curL[r] : c[i > 0 ? i - 1 : 0];
         }
         // file order +x, -x, +y, -y, +z, -z = storage +NzNy, -NzNy, +Nz, -Nz, +1, -1 (axes exchanged: +1, -1, +Nz, -Nz, +NzNy, -NzNy)
         if (fp.swz) {
#pragma unroll
            for (int i = 0; i < V; i++) o[i] = upd7<SG>(a1, a2, c[i], old[r][i], zpv[i], zmv[i], yp[i], ym[i], nxt[r][i], prev[r][i]);
         } else {
#pragma unroll
            for (int i = 0; i < V; i++) o[i] = upd7<SG>(a1, a2, c[i], old[r][i], nxt[r][i], prev[r][i], yp[i], ym[i], zpv[i], zmv[i]);
         }
         if (fp.do_abc) {
            // ABC loss (cpu_engine.h:225-229): u2ba is the old value of the cell, Q from the coordinates
            const int y = y0 + r;
            const int qxy = (qx ? 1 : 0) + ((y == 1 || y == Ny - 2) ? 1 : 0);
            if (qxy > 0 || wave_has_qz) {
#pragma unroll
               for (int i = 0; i < V; i++) {
                  const bool zq = (qzbits >> i) & 1u;
                  if (qxy > 0 || __ballot(zq) != 0ull) {
                     const int Q = qxy + (zq ? 1 : 0);
                     if (Q > 0) {
                        o[i] = abc_loss<SG>(o[i], old[r][i], l * (Real)Q); // (double literal of cpu_engine.h:228 in the CPU-exact mode)
                     }
                  }
               }
            }
         }
#pragma unroll
         for (int i = 0; i < V; i++)
            if ((bits >> i) & 1u) o[i] = old[r][i];
         if (valid[r]) {
            if (NT) __builtin_nontemporal_store(o, (vec *)(po + so[r]));
            else *(vec *)(po + so[r]) = o;
         }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < R; r++) {
         prev[r] = cur[r];
         cur[r] = nxt[r]; curL[r] = nxtL[r]; curR[r] = nxtR[r];
         nxt[r] = nn[r]; nxtL[r] = nnL[r]; nxtR[r] = nnR[r];
         old[r] = oldn[r]; mb[r] = mbn[r];
      }
      hv = hvn;
   }
}


} // namespace pf
