// pf_tb3.h -- temporal blocking, three steps per pass (round 5): k_tb3 produces u^{n+2} and u^{n+3} of the boundary-free box of a
// 7-point room from u^{n-1} and u^n in ONE pass; u^{n+1} never leaves the chip.  16 bytes of compulsory traffic per cell and
// THREE steps (k_tb2_reg: per two steps; a single-step kernel: 12 per step).  On the product path: Engine::step_triple.
//
// Tiling: a workgroup = WT waves stacked in y, each owning R rows x 64 lanes x 16 B (the outermost lanes are z halo -- one 4-cell
// lane per side in fp32, two 2-cell lanes in fp64: 248 / 120 core columns), marching x.  Stage 1 gives u^{n+1} on all WT*R rows of the tile, stage 2 u^{n+2} on
// WT*R - 2, stage 3 u^{n+3} on WT*R - 4: tiles overlap by 4 rows (R = 3, WT = 8: 20 core rows of 24).  The rows a wave needs from its
// neighbours (one above, one below, of the u^n plane that becomes the centre plane next turn and of the u^{n+1} / u^{n+2} planes
// just computed) travel through LDS: every wave publishes its first and last row of the three planes, ONE barrier per plane,
// double-buffered (96 KB at WT = 8: one workgroup per CU, two waves per SIMD).  The two outermost u^n halo rows of a tile have
// no owner in the workgroup and are loaded by the edge waves, a plane ahead.
// Turn x1: stage 1 -> u^{n+1}(x1) from u^n(x1-1, x1, x1+1), u^{n-1}(x1); stage 2 -> u^{n+2}(x1-1) from u^{n+1}(x1-2, x1-1, x1),
// u^n(x1-1); stage 3 -> u^{n+3}(x1-2) from u^{n+2}(x1-3, x1-2, x1-1), u^{n+1}(x1-2).  A chunk [xs, xe) takes turns xs-2 .. xe+1.
// The kernel computes u^{n+1} two cells and u^{n+2} one cell beyond the box: everything within THREE cells of the box must be a
// plain air update (Engine::init_tb2 keeps the box three cells away from boundary nodes, the ABC shell and sources).
// Tiles flagged in the tile list (bit 31: a neighbour of a tile that steps singly) also store their u^{n+1} -- into C, a scratch
// grid --, which the single-step tiles' second step reads.
// Measured on MI355X, 1024^3 fp32 (tools/tb3_probe.py): 3.28-3.30 ms per launch = 1.10 ms per step against 1.53 (k_tb2_reg on
// placed grids) -- 5.0 TB/s of compulsory traffic.
#pragma once
#include "pf_tb2.h"

namespace pf {

constexpr uint32_t TB3_RIM = 0x80000000u; // tile-list flag: store u^{n+1} too

// PROBE: the same code under another name, for the creation-time measurements (as k_tb2_reg's)
// NS = 2: the same tiles, TWO steps -- stages 1 and 2 only, u^{n+1} stored into C by every tile, u^{n+2} into D: what steps the
// last two steps of a run whose length is no multiple of three (Engine::run), with the triples' tile lists and wall regions.
// SRC (round 6; the kernel k_tb3_src below): the tiles within two cells of a SOURCE -- until then three single steps out of memory, the source added between
// them by k_io: a serial tail of four dependent launches behind every triple -- run this form instead, a few workgroups beside the main
// launch: after every stage the samples of the sources the stage has just computed are added in registers, in list order (duplicates
// accumulate as in the reference's serial loop, cpu_engine.h:303-306), before the value is stored, published or used by the next stage.
// Every tile whose computed region reaches a source within the two cells a later stage can still see runs it (Engine::init_tb2_impl marks them),
// so all copies of a halo cell agree.  The main launch's code is untouched (if constexpr), and so is its name.
constexpr int TB3_MAXSRC = 32; // sources k_tb3_src keeps per wave (Engine::init_tb2_impl: more than that, and k_io adds them as before)
template <typename Real, int R, int WT, bool SG, int NS, bool SRC>
__device__ __forceinline__ void tb3_body(const Tb2Params &tp, const Real a1, const Real a2) {
   typedef typename VecOf<Real>::type vec;
   // z halo: a stage loses one cell at each end of a row segment, so three stages need THREE halo cells per side: one 4-cell lane
   // in fp32, two 2-cell lanes in fp64 (k_tb2_reg: two stages, one lane either way)
   constexpr int V = VecOf<Real>::V, W = 64 * V, TRO = WT * R - 4, HL = V >= 3 ? 1 : 2;
   __shared__ __attribute__((aligned(16))) Real sH[2][3][WT][2][W];
   uint32_t b = blockIdx.x;
   int zt, yt, xc;
   bool rim = false;
   if (tp.tiles) {
      const uint32_t t = (uint32_t)tp.tiles[b];
      rim = (t & TB3_RIM) != 0u && tp.C != nullptr;
      b = t & ~TB3_RIM;
      zt = b % tp.nzt; yt = (b / tp.nzt) % tp.nyt; xc = b / (tp.nzt * tp.nyt);
   } else if (tp.band & 1) {
      const uint32_t T = (uint32_t)tp.nzt * tp.nyt, Tp = (T + 7) / 8;
      xc = b / (8 * Tp);
      const uint32_t r = b % (8 * Tp), j = (r % 8) * Tp + r / 8;
      if (j >= T || xc >= tp.nxc) return;
      zt = j % tp.nzt; yt = j / tp.nzt;
   } else { zt = b % tp.nzt; yt = (b / tp.nzt) % tp.nyt; xc = b / (tp.nzt * tp.nyt); }
   const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
   const int ze0 = tp.z_begin - HL * V + zt * (W - 2 * HL * V);
   const int y0 = tp.y_begin - 2 + yt * TRO, yo = y0 + w * R;
   const int xs = tp.x_begin + xc * tp.chunk, xe = min(xs + tp.chunk, tp.x_end);
   const int P = tp.P;
   const int64_t plane = tp.plane;
   const int zc = min(max(ze0 + lane * V, 0), P - V);
   uint32_t off[R];
#pragma unroll
   for (int r = 0; r < R; r++) off[r] = (uint32_t)min(max(yo + r, 0), tp.Ny - 1) * (uint32_t)P + (uint32_t)zc;
   const bool edge_lo = w == 0, edge_hi = w == WT - 1;
   const uint32_t offh = (uint32_t)min(max(edge_lo ? yo - 1 : yo + R, 0), tp.Ny - 1) * (uint32_t)P + (uint32_t)zc; // the tile's outer u^n halo row (edge waves)
   const int z_end = tp.z_end ? tp.z_end : tp.Nz - tp.z_begin, y_end = tp.y_end ? tp.y_end : tp.Ny - tp.y_begin;
   const bool core_col = (lane >= HL && lane <= 63 - HL) && (ze0 + lane * V + V - 1 < z_end);
   bool ok[R];
#pragma unroll
   for (int r = 0; r < R; r++) ok[r] = core_col && (w * R + r >= 2) && (w * R + r <= WT * R - 3) && (yo + r < y_end);
   const Real *A = (const Real *)tp.A, *B = (const Real *)tp.B;
   Real *C = (Real *)tp.C, *D = (Real *)tp.D, *E = (Real *)tp.E;
   auto stencil = [&](const vec &c, const vec &xp, const vec &xm, const vec &yp, const vec &ym, const vec &old) {
      const Real lf = lane_from_lower<true>(c[V - 1]);
      const Real rt = lane_from_upper<true>(c[0]);
      vec o;
#pragma unroll
      for (int i = 0; i < V; i++) {
         const Real zp = (i == V - 1) ? rt : c[i < V - 1 ? i + 1 : V - 1];
         const Real zm = (i == 0) ? lf : c[i > 0 ? i - 1 : 0];
         o[i] = upd7<SG>(a1, a2, c[i], old[i], xp[i], xm[i], yp[i], ym[i], zp, zm);
      }
      return o;
   };
   auto loadrows = [&](const Real *g, int x, vec *d) {
      const Real *pl = g + (int64_t)x * plane;
#pragma unroll
      for (int r = 0; r < R; r++) d[r] = *(const vec *)(pl + off[r]);
   };
   // SRC: the sources whose row one of this wave's rows is, decoded ONCE (list order kept), in LDS: plane, row of the wave, column, place in
   // the list.  (First version: every stage of every turn walked the whole list from memory, two 64-bit divisions per entry -- the few
   // workgroups of this form took 1.04 ms where a workgroup of the main launch takes 0.19, and the main launch beside them 3 % longer.)
   __shared__ int32_t sS[SRC ? WT : 1][SRC ? TB3_MAXSRC : 1][4];
   __shared__ int32_t sNS[SRC ? WT : 1];
   if constexpr (SRC) {
      if (lane == 0) {
         int c = 0;
         for (int j = 0; j < tp.nsrc && c < TB3_MAXSRC; j++) {
            const int64_t ii = tp.src_idx[j];
            const int sx = (int)(ii / plane), rem = (int)(ii % plane), sy = rem / P, sz = rem % P;
            const int rr = sy - yo;
            if (rr < 0 || rr >= R) continue; // (rows are nominal here: a source sits inside the grid, a clamped row is outside it)
            sS[w][c][0] = sx; sS[w][c][1] = rr; sS[w][c][2] = sz; sS[w][c][3] = j;
            c++;
         }
         sNS[w] = c;
      }
      __syncthreads();
   }
   // (the planes this wave's sources lie in, in scalar registers: every other turn's inject is two compares -- walking the list in LDS
   // three times a turn made these few workgroups 0.51 ms long where a workgroup of the main launch takes 0.19)
   int src_lo = 1 << 30, src_hi = -1;
   if constexpr (SRC) {
      for (int c = 0; c < sNS[w]; c++) { src_lo = min(src_lo, sS[w][c][0]); src_hi = max(src_hi, sS[w][c][0]); }
      src_lo = __builtin_amdgcn_readfirstlane(src_lo); src_hi = __builtin_amdgcn_readfirstlane(src_hi);
   }
   // ... and the sample `nn` of those that lie in plane xp added to the rows v[] this lane holds of that plane (nominal, unclamped columns
   // only: a clamped lane is a copy of another cell and feeds nothing valid)
   auto inject = [&](vec(&v)[R], int xp, int64_t nn) {
      if (xp < src_lo || xp > src_hi) return;
      const int cnt = sNS[SRC ? w : 0];
      for (int c = 0; c < cnt; c++) {
         if (sS[SRC ? w : 0][c][0] != xp) continue;
         const int rr = sS[SRC ? w : 0][c][1], dz = sS[SRC ? w : 0][c][2] - (ze0 + lane * V);
         if (dz < 0 || dz >= V || zc != ze0 + lane * V) continue;
         const Real sv = ((const Real *)tp.src_sig)[(int64_t)sS[SRC ? w : 0][c][3] * tp.src_Nt + nn];
#pragma unroll
         for (int r = 0; r < R; r++)
#pragma unroll
            for (int i = 0; i < V; i++)
               if (r == rr && i == dz) v[r][i] += sv;
      }
   };
   vec Bm[R], Bc[R + 2], Bn[R], Bf[R], Ac[R], Af[R];
   vec V1m[R], V1c[R + 2], V1n[R], V2m[R], V2c[R + 2], V2n[R];
   vec BhN = vec{}, BhF = vec{}; // edge waves: the outer halo row of the planes in Bn / Bf
   {
      vec t[R];
      loadrows(B, xs - 3, Bm);
      loadrows(B, xs - 2, t);
#pragma unroll
      for (int r = 0; r < R; r++) Bc[r + 1] = t[r];
      // (prologue only: every wave fetches its two halo rows of the first centre plane itself)
      Bc[0] = *(const vec *)(B + (int64_t)(xs - 2) * plane + (uint32_t)min(max(yo - 1, 0), tp.Ny - 1) * (uint32_t)P + (uint32_t)zc);
      Bc[R + 1] = *(const vec *)(B + (int64_t)(xs - 2) * plane + (uint32_t)min(max(yo + R, 0), tp.Ny - 1) * (uint32_t)P + (uint32_t)zc);
      loadrows(B, xs - 1, Bn);
      loadrows(B, xs, Bf);
      loadrows(A, xs - 2, Ac);
      loadrows(A, xs - 1, Af);
      if (edge_lo || edge_hi) { BhN = *(const vec *)(B + (int64_t)(xs - 1) * plane + offh); BhF = *(const vec *)(B + (int64_t)xs * plane + offh); }
   }
#pragma unroll
   for (int r = 0; r < R; r++) { V1m[r] = vec{}; V2m[r] = vec{}; }
#pragma unroll
   for (int j = 0; j < R + 2; j++) { V1c[j] = vec{}; V2c[j] = vec{}; }
   for (int x1 = xs - 2; x1 <= xe + 1; x1++) {
      const int buf = (x1 - xs) & 1;
      // stage 1: u^{n+1}(x1)
#pragma unroll
      for (int r = 0; r < R; r++) V1n[r] = stencil(Bc[r + 1], Bn[r], Bm[r], Bc[r + 2], Bc[r], Ac[r]);
      if constexpr (SRC) inject(V1n, x1, tp.src_n);
      // (tp.band & 2: slabs of a chain -- the planes beside the box step singly, the box's first and last plane leave their u^{n+1} too)
      const bool xedge = (tp.band & 2) && C != nullptr && (x1 == tp.x_begin || x1 == tp.x_end - 1);
      if ((NS == 2 || rim || xedge) && x1 >= xs && x1 < xe) { // a neighbour of a single-step tile (or the two-step form): its u^{n+1} is needed in memory
         Real *pc = C + (int64_t)x1 * plane;
#pragma unroll
         for (int r = 0; r < R; r++)
            if (ok[r]) __builtin_nontemporal_store(V1n[r], (vec *)(pc + off[r]));
      }
      // stage 2: u^{n+2}(x1-1); its old value is u^n(x1-1)
#pragma unroll
      for (int r = 0; r < R; r++) V2n[r] = stencil(V1c[r + 1], V1n[r], V1m[r], V1c[r + 2], V1c[r], Bm[r]);
      if constexpr (SRC) inject(V2n, x1 - 1, tp.src_n + 1);
      // the planes x1+3 of u^n and x1+2 of u^{n-1}, two turns ahead
      vec Bnew[R], Anew[R], Bhnew = vec{};
      const int xb = min(x1 + 3, xe + 2), xa = min(x1 + 2, xe + 1);
      loadrows(B, xb, Bnew);
      loadrows(A, xa, Anew);
      if (edge_lo || edge_hi) Bhnew = *(const vec *)(B + (int64_t)xb * plane + offh);
      // stage 3: u^{n+3}(x1-2); its old value is u^{n+1}(x1-2)
      if (NS == 3 && x1 - 2 >= xs && x1 - 2 < xe) {
         Real *pe = E + (int64_t)(x1 - 2) * plane;
         if constexpr (SRC) {
            vec o3[R];
#pragma unroll
            for (int r = 0; r < R; r++) o3[r] = stencil(V2c[r + 1], V2n[r], V2m[r], V2c[r + 2], V2c[r], V1m[r]);
            inject(o3, x1 - 2, tp.src_n + 2);
#pragma unroll
            for (int r = 0; r < R; r++)
               if (ok[r]) __builtin_nontemporal_store(o3[r], (vec *)(pe + off[r]));
         } else {
#pragma unroll
            for (int r = 0; r < R; r++) {
               const vec o = stencil(V2c[r + 1], V2n[r], V2m[r], V2c[r + 2], V2c[r], V1m[r]);
               if (ok[r]) __builtin_nontemporal_store(o, (vec *)(pe + off[r]));
            }
         }
      }
      if (x1 - 1 >= xs && x1 - 1 < xe) {
         Real *pd = D + (int64_t)(x1 - 1) * plane;
#pragma unroll
         for (int r = 0; r < R; r++)
            if (ok[r]) __builtin_nontemporal_store(V2n[r], (vec *)(pd + off[r]));
      }
      // publish the rows the neighbouring waves need next turn
      *(vec *)&sH[buf][0][w][0][lane * V] = Bn[0];  *(vec *)&sH[buf][0][w][1][lane * V] = Bn[R - 1];
      *(vec *)&sH[buf][1][w][0][lane * V] = V1n[0]; *(vec *)&sH[buf][1][w][1][lane * V] = V1n[R - 1];
      if (NS == 3) { *(vec *)&sH[buf][2][w][0][lane * V] = V2n[0]; *(vec *)&sH[buf][2][w][1][lane * V] = V2n[R - 1]; }
      __syncthreads();
      vec hB0, hB1, hV10 = vec{}, hV11 = vec{}, hV20 = vec{}, hV21 = vec{};
      if (!edge_lo) { hB0 = *(const vec *)&sH[buf][0][w - 1][1][lane * V]; hV10 = *(const vec *)&sH[buf][1][w - 1][1][lane * V]; if (NS == 3) hV20 = *(const vec *)&sH[buf][2][w - 1][1][lane * V]; }
      else hB0 = BhN;
      if (!edge_hi) { hB1 = *(const vec *)&sH[buf][0][w + 1][0][lane * V]; hV11 = *(const vec *)&sH[buf][1][w + 1][0][lane * V]; if (NS == 3) hV21 = *(const vec *)&sH[buf][2][w + 1][0][lane * V]; }
      else hB1 = BhN;
      // rotate
#pragma unroll
      for (int r = 0; r < R; r++) {
         Bm[r] = Bc[r + 1]; Bc[r + 1] = Bn[r]; Bn[r] = Bf[r]; Bf[r] = Bnew[r];
         Ac[r] = Af[r]; Af[r] = Anew[r];
         V1m[r] = V1c[r + 1]; V1c[r + 1] = V1n[r];
         V2m[r] = V2c[r + 1]; V2c[r + 1] = V2n[r];
      }
      Bc[0] = hB0; Bc[R + 1] = hB1; V1c[0] = hV10; V1c[R + 1] = hV11; V2c[0] = hV20; V2c[R + 1] = hV21;
      BhN = BhF; BhF = Bhnew;
   }
}

template <typename Real, int R, int WT, bool SG = false, bool PROBE = false, int NS = 3>
__global__ __launch_bounds__(64 * WT) void k_tb3(Tb2Params tp, Real a1, Real a2) { tb3_body<Real, R, WT, SG, NS, false>(tp, a1, a2); }
// the tiles within two cells of a source (Engine::launch_tb3_src), beside the main launch
template <typename Real, int R, int WT, bool SG = false, int NS = 3>
__global__ __launch_bounds__(64 * WT) void k_tb3_src(Tb2Params tp, Real a1, Real a2) { tb3_body<Real, R, WT, SG, NS, true>(tp, a1, a2); }

} // namespace pf
