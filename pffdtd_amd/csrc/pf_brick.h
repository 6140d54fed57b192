// pf_brick.h -- the FRAME of a blocked pair / triple: where two walls meet (round 6; 7-point, CPU-exact or safeguarded arithmetic).
//
// The wall regions of pf_wall.h are fast where all pencils of a block look alike -- a wall away from edges and corners.  Along the
// twelve edges of a box room a pencil lies INSIDE the other wall's layer (every cell a node), lanes are ghost or ABC cells, march
// planes mirror: until round 5 the "generic" blocks of k_wall2 carried those, 12 % of the blocks and a third of the shell's
// SIMD-time (profiles/r05_wall_counters.txt), and -- being unable to step three times in one pass (a halo node's branch state after
// the first step has nowhere to live) -- they kept the whole shell at two steps + one.
//
// Here the frame is cut into BRICKS: small boxes (a few cells across, 16 along the edge), each stepped by one workgroup with the
// brick and a halo of `ns` cells held in LDS: `ns` plain single steps of the reference's loop, one after the other, on a region that
// shrinks by one cell per step -- the cells a step can no longer compute are exactly the halo the next one does not need.  No
// pencils, no pipelines: every cell is generic (air + ABC loss by its coordinates, rigid node, frequency-dependent node) and the
// volume is tiny (0.2 % of a 1024^3 grid's cells), so what matters is that a brick is self-contained: it reads u^{n-1}, u^n and
// the old branch state only, like a wall region, and is independent of every other launch of the pass.
//
//   * ghost cells are never stored or loaded: a cell at index 1 (N-2) takes its +1 (-1) neighbour for the missing one (the flip of
//     cpu_engine.h:145-172 mirrors the cell two further in; a 7-point update reads face ghosts only);
//   * per cell of the extended box one info word (adjacency bits, node / frequency-dependent flags, the ABC count Q of an air
//     cell), built by the host; per brick the list of its frequency-dependent nodes: after the cell pass of a step, thread t takes
//     nodes t, t + 256, ... with their branch state in registers for all `ns` steps -- the ODEs run dense whatever the geometry;
//   * branch state: read from sv_in, the state after the last step written to sv_out for the nodes the brick owns (the wall
//     regions' double buffer); node values of step s go to O[s] (the u2b rotation of cpu_engine.h:290-301 for the other paths).
//
// Arithmetic: upd7 / upd_rigid / abc_loss of pf_kernels.h, fd_regs of pf_wall.h, neighbours in file order -- bit-identical to
// cpu_engine.h:175-194,225-229,234-257,290-301,363-405.
#pragma once
#include "pf_wall.h"

namespace pf {

constexpr int BRICK_T = 256;  // threads per brick
constexpr int BRICK_KN = 2;   // frequency-dependent nodes a thread carries at most
constexpr int BRICK_HALO = 3; // cells of halo a brick is built with (= the most steps one launch may take)

struct Brick {
   int32_t e0[3], en[3];  // extended box: first cell (x, y, z; >= 1) and extents (ghost cells are never part of it)
   int32_t o0[3], o1[3];  // the cells it owns (stores): [o0, o1)
   uint32_t info_off;     // its first info word
   uint32_t los_off, nlos; // its frequency-dependent nodes
};

template <typename Real> struct BrickParams {
   const Real *A, *B;     // u^{n-1}, u^n
   Real *G[3];            // where steps 1 .. ns go
   Real *O[3];            // node values of those steps (lossy arrays' order)
   int64_t plane;
   int32_t Nx, Ny, Nz, P;
   const Brick *brk;
   const uint32_t *info;  // adjacency bits | 0x40 node | 0x80 frequency-dependent; air cells: ABC count Q << 8
   const uint2 *los;      // .x = cell of the extended box | owned << 31, .y = position in the lossy arrays
   const Real *x2, *x1;   // node values u^{n-1}, u^n (lossy arrays' order): the u2b of steps 1 and 2 (cpu_engine.h:290-301); nobody writes them
                          // during the pass (single domains rotate FIVE node-value buffers: Engine::step_triple)
   const Real *sv_in, *sg_in;
   Real *sv_out, *sg_out;
   const Real *ssaf;
   const int8_t *mat, *Mb;
   const MatQuadT<Real> *mq;
   const Real *beta;
   Real lo2, sl2, l;
   int32_t nmat, ns;
   int32_t first, last;   // does this grid hold the ghost plane at x = 0 / x = Nx - 1?  (a slab of a chain: only the chain's ends do; plane 0 /
                          // Nx - 1 of an inner cut is the neighbour's data, which a brick never needs: its bars keep three planes from the cuts)
};

template <typename Real> struct BrickLds { // (what fd_regs wants of WallLds, carved from the dynamic allocation: nmat materials, not 64)
   const MatQuadT<Real> *mq;
   const Real *beta;
   const int32_t *M;
};
// bytes of LDS a brick of `cells` extended cells needs
template <typename Real> __host__ __device__ inline size_t brick_lds_bytes(int64_t cells, int nmat) {
   size_t b = (size_t)nmat * 12 * sizeof(MatQuadT<Real>) + (size_t)nmat * sizeof(Real) + (size_t)nmat * sizeof(int32_t);
   b = (b + 15) & ~(size_t)15;
   return b + 3 * (size_t)cells * sizeof(Real);
}

template <typename Real, int MC, bool SG>
__global__ __launch_bounds__(BRICK_T) void k_brick(BrickParams<Real> bp, Real a1, Real a2) {
   extern __shared__ __attribute__((aligned(16))) unsigned char brick_smem[];
   const Brick bk = bp.brk[blockIdx.x];
   const int tid = threadIdx.x;
   const uint32_t ex = (uint32_t)bk.en[0], ey = (uint32_t)bk.en[1], ez = (uint32_t)bk.en[2];
   const uint32_t ncell = ex * ey * ez, syx = ey * ez;
   MatQuadT<Real> *lmq = (MatQuadT<Real> *)brick_smem;
   Real *lbeta = (Real *)(lmq + bp.nmat * 12);
   int32_t *lM = (int32_t *)(lbeta + bp.nmat);
   Real *uo = (Real *)(((uintptr_t)(lM + bp.nmat) + 15) & ~(uintptr_t)15), *uc = uo + ncell, *un = uc + ncell;
   for (int i = tid; i < bp.nmat * 12; i += BRICK_T) lmq[i] = bp.mq[i];
   for (int i = tid; i < bp.nmat; i += BRICK_T) { lbeta[i] = bp.beta[i]; lM[i] = bp.Mb[i]; }
   const BrickLds<Real> lds{lmq, lbeta, lM};
   // this thread's frequency-dependent nodes: state and parameters, for all the steps
   Real fv[BRICK_KN][12], fg[BRICK_KN][12], fsf[BRICK_KN], fx2[BRICK_KN], fx1[BRICK_KN];
   int32_t fk[BRICK_KN], fli[BRICK_KN];
   uint32_t fc[BRICK_KN];
#pragma unroll
   for (int k = 0; k < BRICK_KN; k++) {
      const uint32_t j = (uint32_t)tid + (uint32_t)k * BRICK_T;
      fc[k] = 0xffffffffu; fli[k] = 0; fk[k] = 0; fsf[k] = Real(0); fx2[k] = Real(0); fx1[k] = Real(0);
#pragma unroll
      for (int m = 0; m < 12; m++) { fv[k][m] = Real(0); fg[k][m] = Real(0); }
      if (j < bk.nlos) {
         const uint2 e = bp.los[bk.los_off + j];
         fc[k] = e.x; fli[k] = (int32_t)e.y;
#pragma unroll
         for (int m = 0; m < 12; m++)
            if (m < MC) { fv[k][m] = bp.sv_in[st_idx(m, fli[k])]; fg[k][m] = bp.sg_in[st_idx(m, fli[k])]; }
         fsf[k] = bp.ssaf[fli[k]];
         fk[k] = bp.mat[fli[k]];
         fx2[k] = bp.x2[fli[k]];
         fx1[k] = bp.x1[fli[k]];
      }
   }
   // u^{n-1}, u^n of the extended box
   for (uint32_t idx = tid; idx < ncell; idx += BRICK_T) {
      const uint32_t iz = idx % ez, t = idx / ez, iy = t % ey, ix = t / ey;
      const int64_t a = (int64_t)(bk.e0[0] + (int)ix) * bp.plane + (int64_t)(bk.e0[1] + (int)iy) * bp.P + (bk.e0[2] + (int)iz);
      uo[idx] = bp.A[a];
      uc[idx] = bp.B[a];
   }
   __syncthreads();
   const int N[3] = {bp.Nx, bp.Ny, bp.Nz};
   for (int s = 1; s <= bp.ns; s++) {
      // what this step can compute: `s` cells off every face of the extended box -- but a face that ends at the grid's own shell
      // (index 1 / N-2: beyond it only the mirrored ghost cell) loses nothing
      int lo[3], hi[3];
#pragma unroll
      for (int d = 0; d < 3; d++) {
         const bool glo = d > 0 || bp.first, ghi = d > 0 || bp.last; // is index 1 / N-2 of this axis the grid's own shell?
         lo[d] = (bk.e0[d] > 1 || !glo) ? s : 0;
         hi[d] = (bk.e0[d] + bk.en[d] < N[d] - 1 || !ghi) ? bk.en[d] - s : bk.en[d];
      }
      for (uint32_t idx = tid; idx < ncell; idx += BRICK_T) {
         const uint32_t iz = idx % ez, t = idx / ez, iy = t % ey, ix = t / ey;
         if ((int)ix < lo[0] || (int)ix >= hi[0] || (int)iy < lo[1] || (int)iy >= hi[1] || (int)iz < lo[2] || (int)iz >= hi[2]) continue;
         const int gx = bk.e0[0] + (int)ix, gy = bk.e0[1] + (int)iy, gz = bk.e0[2] + (int)iz;
         const uint32_t w = bp.info[bk.info_off + idx];
         const Real c = uc[idx], old = uo[idx];
         // ghost cells mirror the cell two further in (cpu_engine.h:145-172): at index 1 the -1 neighbour IS the +1 neighbour
         const Real xp = uc[(gx == bp.Nx - 2 && bp.last) ? idx - syx : idx + syx], xm = uc[(gx == 1 && bp.first) ? idx + syx : idx - syx];
         const Real yp = uc[gy == bp.Ny - 2 ? idx - ez : idx + ez], ym = uc[gy == 1 ? idx + ez : idx - ez];
         const Real zp = uc[gz == bp.Nz - 2 ? idx - 1 : idx + 1], zm = uc[gz == 1 ? idx + 1 : idx - 1];
         Real p;
         if (w & 0x40u) { // boundary node (cpu_engine.h:234-257)
            const Real nb[6] = {xp, xm, yp, ym, zp, zm};
            p = upd_rigid<SG, 6>(a2, bp.sl2, w & 63u, c, old, nb);
         } else {
            p = upd7<SG>(a1, a2, c, old, xp, xm, yp, ym, zp, zm); // (cpu_engine.h:175-194)
            const uint32_t Q = (w >> 8) & 3u;
            if (Q) p = abc_loss<SG>(p, old, bp.l * (Real)Q); // (cpu_engine.h:225-229)
         }
         un[idx] = p;
      }
      __syncthreads();
      // the branch ODEs of the frequency-dependent nodes, dense (cpu_engine.h:290-301, 363-405); a node the step could not compute
      // (halo) carries garbage from here on, which nothing valid ever reads
#pragma unroll
      for (int k = 0; k < BRICK_KN; k++) {
         if (fc[k] != 0xffffffffu) {
            const uint32_t cell = fc[k] & 0x7fffffffu;
            // the node's value two steps back: the engine's node-value buffers for steps 1 and 2 (what every other path reads: they
            // equal the grid's cells unless a caller planted fields with pf_engine_set_grid), the brick's own step 1 afterwards
            const Real u2 = s == 1 ? fx2[k] : (s == 2 ? fx1[k] : uo[cell]);
            const Real u = fd_regs<Real, MC>(un[cell], u2, fsf[k], fk[k], fv[k], fg[k], fv[k], fg[k], lds, bp.lo2);
            un[cell] = u;
            if (fc[k] >> 31) bp.O[s - 1][fli[k]] = u;
         }
      }
      __syncthreads();
      { // the owned cells of this step
         Real *G = bp.G[s - 1];
         const uint32_t oy = (uint32_t)(bk.o1[1] - bk.o0[1]), oz = (uint32_t)(bk.o1[2] - bk.o0[2]), nown = (uint32_t)(bk.o1[0] - bk.o0[0]) * oy * oz;
         for (uint32_t j = tid; j < nown; j += BRICK_T) {
            const uint32_t kz = j % oz, t = j / oz, ky = t % oy, kx = t / oy;
            const int gx = bk.o0[0] + (int)kx, gy = bk.o0[1] + (int)ky, gz = bk.o0[2] + (int)kz;
            const uint32_t idx = ((uint32_t)(gx - bk.e0[0]) * ey + (uint32_t)(gy - bk.e0[1])) * ez + (uint32_t)(gz - bk.e0[2]);
            G[(int64_t)gx * bp.plane + (int64_t)gy * bp.P + gz] = un[idx];
         }
      }
      Real *t = uo; uo = uc; uc = un; un = t;
      // (no barrier: the next step writes what was `uo`, last read before the barrier above; the stores read what is `uc` now)
   }
#pragma unroll
   for (int k = 0; k < BRICK_KN; k++) {
      if (fc[k] != 0xffffffffu && (fc[k] >> 31)) {
#pragma unroll
         for (int m = 0; m < 12; m++)
            if (m < MC) { bp.sv_out[st_idx(m, fli[k])] = fv[k][m]; bp.sg_out[st_idx(m, fli[k])] = fg[k][m]; }
      }
   }
}

} // namespace pf
