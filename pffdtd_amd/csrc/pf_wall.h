// pf_wall.h -- the SHELL of a temporally blocked pair stepped in pairs too (7-point; CPU-exact or GPU-safeguarded arithmetic).
//
// k_tb2_reg (pf_tb2.h) advances the boundary-free box by two steps per pass; until round 3 everything around it -- the wall
// layers with their frequency-dependent nodes, the ABC cells, the ghost mirrors -- was stepped twice by single-step kernels:
// ~12 launches per pair, every branch-ODE state (vh1, gh1: 176 B per node at Mb = 11) read and written once per STEP, the
// column strips' half-used lines fetched once per step.  k_wall2 does both steps of a wall region in one pass:
//
//   * a region = the cells between a grid face and the box ("pencils" of DP cells along the face normal, pencil cell 0 /
//     DP-1 = the ghost cell), tiled along a "lane" axis (64 lanes, one pencil each, 60 of them owned) and marched along the
//     third axis in chunks.  Regions normal to x and y have their lanes along z (unit stride: 256-byte runs per pencil cell),
//     the regions normal to z have their lanes along y and load their pencils as 16-byte vectors (one 128-byte line per row
//     and side, read once per pair for u^n and u^{n-1}, written once for u^{n+1} and u^{n+2}).
//   * per march step m: stage 1 = u^{n+1}(m) on the pencil cells 1 .. DP-2 from u^n(m-1 .. m+1) and u^{n-1}(m); stage 2 =
//     u^{n+2}(m-1) on the OWNED cells from u^{n+1}(m-2 .. m) and u^n(m-1) -- the scheme of k_tb2_reg, but every cell is
//     generic: air (+ ABC loss by its coordinates), ghost (mirrored in registers: normal axis; by a wave shuffle: lane axis;
//     by taking the other march plane: march axis), rigid boundary node, frequency-dependent boundary node.  Lane-axis
//     neighbours come from the DPP wave shifts, so lanes 0 / 63 are halo (stage 1 invalid), lanes 1 / 62 are stage-1-only.
//   * nodes: per pencil a 16-byte entry -- a mask of its node cells, the adjacency bits and frequency-dependent flags of its
//     first five nodes, the position of its first frequency-dependent node in the lossy arrays -- plus a record list for
//     whatever does not fit (edges, corners).  The wave walks the union of its lanes' masks (box rooms: two turns, one for
//     the rigid layer, one for the lossy layer, every lane busy), so the branch ODEs run dense.
//   * software pipeline: the loads of march step m+1 (u^n, u^{n-1}, the entry of m+2, the branch state of the first
//     frequency-dependent node of pencil m+1) are issued at the top of step m -- a wave's march is a chain of dependent
//     steps and only two or three waves fit a SIMD, so nothing else hides the latency (first version, everything loaded
//     where it was needed: 9 dependent round trips per step, 5x slower).  That node's state stays in registers between its
//     stage 1 (step m) and its stage 2 (step m+1) and is stored once.
//   * everything a region needs beyond its own cells (one cell of halo in every direction, incl. the branch state of the
//     nodes there) is RECOMPUTED from u^n / u^{n-1}, never read from the grids being written: the regions are independent
//     of each other, of the box kernel and of the launch order.  For that the branch state is double-buffered (read
//     sv_in / sg_in, write sv_out / sg_out; the engine swaps after the pair) and the node values of the two steps go to
//     buffers nobody reads during the pair (o1, o2; x2 = u^{n-1} of the nodes is only read).
//   * round 6: blocks whose pencils are not alike -- the frame of the shell: edges, corners -- are no longer this kernel's in single
//     domains (pf_brick.h steps them in LDS); what remains is alike and may take THREE steps per pass (NS = 3, below: three lanes /
//     march planes / pencil cells of halo instead of two), with the pencil's geometry compiled in where it is the standard one
//     (GD, HI).  The generic path stays for slabs of a chain, rooms that are no plain box and the tests.
//
// Arithmetic: upd7 / upd_rigid / abc_loss of pf_kernels.h and the branch ODEs in fd_core's order, neighbours in FILE order
// whatever the pencil's orientation -- bit-identical to the single-step kernels and to
// cpu_engine.h:175-194,225-229,234-257,290-301,363-405.
#pragma once
#include <type_traits>
#include "pf_kernels.h"

namespace pf {

constexpr int WALL_MAXREG = 6;
constexpr int WALL_LT = 60; // owned lanes per tile of the two-step tables (lanes 2 .. 61; three-step tables: 58, WallRegion::lt)
#define PF_WALL_MAXMAT 64     // (= PF_MNM, fdtd_data.h:35)

struct WallRegion {
   int32_t mode;        // pencils along 0: x (lanes z, march y), 1: y (lanes z, march x), 2: z (lanes y, march x; vector loads)
   int32_t nbase;       // normal coordinate of pencil cell 0
   int32_t kg;          // pencil index of the ghost cell (mirrors the cell two inside), -1: none
   int32_t ko0, ko1;    // pencil cells this region owns (writes): [ko0, ko1)
   int32_t kb0, kb1;    // pencil cells of u^n / u^{n-1} it loads: [kb0, kb1)
   int32_t l0, l1;      // lane-axis coordinates owned: [l0, l1)
   int32_t m0, m1;      // march coordinates owned: [m0, m1)
   int32_t mchunk, nlt; // march steps per block, lane tiles
   int32_t nlp;         // pencils per march step in the pencil table (nlt * lt + 2 * hl)
   uint32_t blk0;       // first block of the region in the launch
   int32_t hl, lt, hm;  // halo lanes per side of a tile (2; 3 where the region may take THREE steps per pass), owned lanes per tile (64 - 2 hl),
                        // march planes the pencil table holds before m0 (1 / 2): it covers march steps m0 - hm .. m1 + hm - 1
   int64_t pen_off;     // the region's pencil table: entry (m - (m0 - hm)) * nlp + (lc - (l0 - hl))
};

// pencil entry: .x bit k = pencil cell k is a boundary node; .y = index of its first record | pencil cell of its first
// frequency-dependent node << 27; .z = adjacency bits of its first five nodes (6 each); .w = frequency-dependent flags of
// those five | position of the first frequency-dependent one in the lossy arrays << 8
// record (all nodes of a pencil, in pencil order): adjacency bits | 0x40 frequency-dependent | position in the lossy arrays << 8
template <typename Real> struct WallParams {
   const Real *A, *B;       // u^{n-1}, u^n
   Real *C, *D, *E;         // u^{n+1}, u^{n+2}, u^{n+3} (NS = 3)
   int64_t plane;
   int32_t Nx, Ny, Nz, P, first, last;
   int32_t nreg;
   WallRegion reg[WALL_MAXREG];
   const uint4 *pen;
   const uint32_t *rec;
   const uint4 *blk;        // the launch's blocks: .x = region | lane tile << 3 | march chunk << 16; FAST: .y / .z = node mask / adjacency words
                            // common to the block's pencils, .w = lossy flags | pencil cell of the frequency-dependent node << 8
   const Real *sv_in, *sg_in;
   Real *sv_out, *sg_out;   // branch state vh1 / gh1 before and after the pair (64-node blocks, st_idx)
   const Real *x2, *x1;     // node values u^{n-1} (step 1) and u^n (step 2): the u2b of cpu_engine.h:290-301
   Real *o1, *o2, *o3;      // node values u^{n+1}, u^{n+2} (, u^{n+3}).  Single domains: buffers nobody reads during the pass (round 6:
                            // five node-value buffers); slabs of a chain: o2 == x1 (an owner reads its node's u^n before it stores its u^{n+2})
   const Real *ssaf;
   const int8_t *mat, *Mb;
   const MatQuadT<Real> *mq;
   const Real *beta;
   Real lo2, sl2, l;
   int32_t mmax, nmat;
};

// (each element passes through an empty asm: otherwise the compiler turns the chain of selects over array elements into ONE
// load with a computed address, which keeps the whole pencil array in scratch memory -- 176 / 576 bytes per lane, 5x slower)
template <typename Real, int N> __device__ __forceinline__ Real wall_sel(const Real (&a)[N], int k) { // k wave-uniform
   Real r = a[0];
#pragma unroll
   for (int i = 1; i < N; i++) {
      Real t = a[i];
      asm("" : "+v"(t));
      r = (k == i) ? t : r;
   }
   return r;
}

// The branch ODEs of one node with its state in registers (fd_core's arithmetic, cpu_engine.h:363-405).  The materials'
// coefficients come from a copy in LDS (scalar loads of them serialised: two dozen dependent waits per node).
template <typename Real> struct WallLds {
   MatQuadT<Real> mq[PF_WALL_MAXMAT * 12];
   Real beta[PF_WALL_MAXMAT];
   int32_t M[PF_WALL_MAXMAT];
};
// Generic blocks: the frequency-dependent nodes beyond a pencil's first one go through memory (fd_core).  Where two walls meet, ONE
// lane of a tile holds a pencil that lies inside the other wall's layer -- every cell a node -- and the wave used to run fd_core once
// per pencil cell for that single lane (6 turns x 2 stages per march step: three quarters of a generic block's instructions).  The
// jobs are parked here instead, [lane][pencil cell], and run TRANSPOSED after the node loop: lane L takes job slot t * 64 + L of the
// flattened array, so the six jobs of one pencil run side by side in one pass.
template <typename Real> struct WallJobs {
   Real p[64 * 8];
   int32_t li[64 * 8]; // position in the lossy arrays | owner << 30
   uint32_t mask[64];  // pencil cells with a job, per lane
};
template <typename Real, int mmax, typename LDS = WallLds<Real>> // (LDS: anything with mq[], beta[], M[] -- pf_brick.h carves its own)
__device__ __forceinline__ Real fd_regs(Real p, Real u2, Real sf, int32_t k, const Real (&v1)[12], const Real (&g1)[12], Real (&v1o)[12], Real (&g1o)[12],
                                        const LDS &L, Real lo2) {
   // mmax = the largest branch count of the scene (uniform).  Branches m >= the node's own count M are computed and thrown away
   // (selects): a per-lane branch around every m put each LDS read of a coefficient and its wait into a block of its own --
   // two dozen dependent LDS round trips per node, with one wave per SIMD nothing to hide them.
   const int M = L.M[k];
   const Real two = 2.0, one = 1.0;
   const Real g = lo2 * sf * L.beta[k];
   MatQuadT<Real> q[12];
#pragma unroll
   for (int m = 0; m < 12; m++)
      if (m < mmax) q[m] = L.mq[k * 12 + m];
   const Real fac = two * lo2 * sf / (one + g);
   Real u = p;
   u = (u + g * u2) / (one + g);
#pragma unroll
   for (int m = 0; m < 12; m++) {
      if (m < mmax) {
         const Real t = u - fac * (two * q[m].bDh * v1[m] - q[m].bFh * g1[m]);
         u = (m < M) ? t : u;
      }
   }
   const Real du = u - u2;
#pragma unroll
   for (int m = 0; m < 12; m++) {
      if (m < mmax) {
         const Real v0 = q[m].b * du + q[m].bd * v1[m] - two * q[m].bFh * g1[m];
         const Real gn = g1[m] + (v0 + v1[m]) / two;
         g1o[m] = (m < M) ? gn : g1[m];
         v1o[m] = (m < M) ? v0 : v1[m];
      } else { g1o[m] = g1[m]; v1o[m] = v1[m]; }
   }
   return u;
}

// FAST: the host found every pencil the block evaluates (all lanes, all march steps) to have the SAME structure -- the same node
// cells with the same adjacency, at most one frequency-dependent node -- and no ghost or ABC cell along the lane and march
// axes: walls away from edges and corners, the bulk of the work.  The structure then arrives with the block (ds*) and the
// per-lane node decoding, the node loop, the lane shuffles and the march-axis mirrors are compiled out.
// NODES = false (FAST only): none of the block's pencils holds a boundary node (the plain-air part of a wide column strip).
// NS = 3 (round 6; alike blocks only): THREE steps in one pass -- stage 3 = u^{n+3}(m-2) from u^{n+2}(m-3 .. m-1) and u^{n+1}(m-2); one more
// lane, march plane and pencil cell of halo on every side (region tables built with hl = 3, hm = 2), the branch state read and written ONCE
// per triple instead of twice, u^{n+1} / u^{n+2} of the region never re-read.  A node's u2b is x2 at stage 1, x1 at stage 2, and its own
// stage-1 value at stage 3.
// NS = 1: ONE step of the region (round 5: the third step of a triple) -- stage 1 alone, with the owned cells, the node value and the
// branch state stored after it.  No halo is needed then (what stage 1 computes outside the owned cells is thrown away), so the state
// may be updated in place (sv_in == sv_out) and the node values go from x2 to o1.
// GD > 0 (round 6; alike blocks of three-step launches): the pencil's geometry is the STANDARD one of a box whose margin on the pencil axis is GD
// cells -- low side (HI = false): ghost cell 0, owned cells 1 .. GD-1, pencil from coordinate 0; high side: ghost cell DP-1, owned cells
// DP-GD .. DP-2, pencil from N - DP -- and a compile-time constant: the ghost mirror, the owned range of every store, the ABC cell and the load
// clamps cost no scalar compares and selects per cell and stage any more (x / y regions: -12 % vector, -29 % scalar instructions per march
// step).  The host launches these bodies only when every region of the launch has that geometry (Engine::launch_walls_x), else GD = 0.
template <typename Real, int DP, int MODE, bool FAST, bool NODES, int MC, bool SG, int NS = 2, int GD = 0, bool HI = false>
__device__ __forceinline__ void wall_body(const WallParams<Real> &wp, const WallRegion &R, const int j, const int c, const Real a1, const Real a2,
                                          const WallLds<Real> *ldsp, const uint32_t dsx, const uint32_t dsz, const uint32_t dsw, WallJobs<Real> *jobs = nullptr) {
   constexpr bool VEC = MODE == 2;
   typedef typename VecOf<Real>::type vec;
   constexpr int V = VecOf<Real>::V;
   static_assert(!VEC || DP % V == 0, "vector pencils hold whole vectors");
   const int lane = threadIdx.x;
   constexpr bool mx = MODE != 0;                                  // the march axis is x
   const int NL = VEC ? wp.Ny : wp.Nz;                             // extent of the lane axis
   const int NM = mx ? wp.Nx : wp.Ny;                              // ... of the march axis
   const int NN = VEC ? wp.Nz : (MODE == 1 ? wp.Ny : wp.Nx);       // ... of the pencil axis
   const bool mg_lo = mx ? (wp.first != 0) : true, mg_hi = mx ? (wp.last != 0) : true; // ghost planes at the ends of the march axis?
   const bool ng_lo = MODE == 0 ? (wp.first != 0) : true, ng_hi = MODE == 0 ? (wp.last != 0) : true;
   const int64_t sl = VEC ? (int64_t)wp.P : 1, sm = mx ? wp.plane : (int64_t)wp.P;
   const int64_t sn = VEC ? 1 : (MODE == 1 ? (int64_t)wp.P : wp.plane);
   static_assert(NS <= 2 || FAST, "three steps per pass: alike blocks only");
   const int hl = R.hl, lt = R.lt;
   const int lc = R.l0 - hl + lt * j + lane;                       // this lane's coordinate on the lane axis
   int lsrc = min(max(lc, 0), NL - 1);                             // where its u^n comes from: ghost cells mirror
   if (lsrc == 0) lsrc = 2;
   else if (lsrc == NL - 1) lsrc = NL - 3;
   const bool lg_lo = lc == 0, lg_hi = lc == NL - 1;
   const bool tile_lg = !FAST && __ballot(lg_lo || lg_hi) != 0ull;
   const bool own_lane = lane >= hl && lane <= 63 - hl && lc < R.l1;
   const bool eval_lane = lane >= 1 && lane <= 62 && lc <= R.l1 + (NS == 3 ? 1 : 0); // stage 1 is valid (and needed) here
   const int ms = R.m0 + c * R.mchunk, me = min(ms + R.mchunk, R.m1);
   if (ms >= me) return;
   const WallLds<Real> &lds = *ldsp;
   // Wave-uniform values the loop body branches on.  They are made opaque once per march step (below): left alone, the compiler
   // hoists every uniform predicate derived from them out of the march loop, runs out of scalar registers and spills them into
   // vector-register lanes -- 300-500 v_readlane / v_writelane per step, a fifth of the vector instructions.
   constexpr bool CG = GD > 0; // constant pencil geometry
   static_assert(!CG || (FAST && NS == 3 && GD + 3 <= DP), "constant pencil geometry: alike blocks of three-step launches");
   int rkg = CG ? (HI ? DP - 1 : 0) : R.kg, rko0 = CG ? (HI ? DP - GD : 1) : R.ko0, rko1 = CG ? (HI ? DP - 1 : GD) : R.ko1;
   int rkb0 = CG ? (HI ? DP - GD - 3 : 0) : R.kb0, rkb1 = CG ? (HI ? DP : GD + 3) : R.kb1, rnbase = CG ? (HI ? NN - DP : 0) : R.nbase;
   uint32_t usx = dsx, usz = dsz, usw = dsw;
   auto msrc = [&](int m) __attribute__((always_inline)) {
      m = min(max(m, 0), NM - 1);
      if (FAST) return m;
      if (m == 0 && mg_lo) return 2;
      if (m == NM - 1 && mg_hi) return NM - 3;
      return m;
   };
   const int64_t lbase = (int64_t)lsrc * sl + (int64_t)rnbase * sn;
   const int ql = (lc == 1 || lc == NL - 2) ? 1 : 0;
   const bool tile_ql = !FAST && __ballot(ql != 0) != 0ull;

   // Loads are unconditional (a cell outside [kb0, kb1) re-reads the nearest one inside) and nothing touches what they return
   // before the end of the march step: the mirror of the ghost cell and the masking of the entries happen at the rotation.
   auto load_pencil = [&](const Real *G, int m, Real(&b)[DP]) __attribute__((always_inline)) {
      const Real *pl = G + (int64_t)msrc(m) * sm + lbase;
      if constexpr (VEC) {
#pragma unroll
         for (int v = 0; v < DP / V; v++) {
            const vec t = *(const vec *)(pl + v * V);
#pragma unroll
            for (int i = 0; i < V; i++) b[v * V + i] = t[i];
         }
      } else {
#pragma unroll
         for (int k = 0; k < DP; k++) b[k] = pl[(int64_t)min(max(k, rkb0), rkb1 - 1) * sn];
      }
   };
   auto mirror = [&](Real(&b)[DP]) __attribute__((always_inline)) { // the ghost cell of a pencil = the cell two inside
      // (selects, not branches: a wave-uniform `if` per cell is a compare + branch around one move, three pencils per march step)
#pragma unroll
      for (int k = DP - 1; k >= 0; k--) {
         const Real src = (k == 0) ? b[2] : b[k >= 2 ? k - 2 : 0];
         b[k] = (k == rkg) ? src : b[k];
      }
   };
   auto load_ent = [&](int m) __attribute__((always_inline)) { // march steps R.m0 - hm .. R.m1 + hm - 1 have entries
      const uint4 *e = wp.pen + (R.pen_off + (int64_t)(min(m, R.m1 + R.hm - 1) - (R.m0 - R.hm)) * R.nlp + (lt * j + lane));
      if (FAST && !NODES) return make_uint4(0u, 0u, 0u, 0u);
      if (FAST) return make_uint4(0u, 0u, 0u, e->w); // (only the place of the frequency-dependent node differs from lane to lane)
      return *e;
   };
   auto mask_ent = [&](uint4 e, int m) __attribute__((always_inline)) {
      if (!eval_lane || m > R.m1 + (NS == 3 ? 1 : 0)) { e.x = 0u; e.w = 0u; }
      return e;
   };
   auto store_pencil = [&](Real *G, int m, const Real(&v)[DP]) __attribute__((always_inline)) {
      if (!own_lane) return;
      Real *pl = G + (int64_t)m * sm + (int64_t)lc * sl + (int64_t)rnbase * sn;
      if constexpr (VEC) {
#pragma unroll
         for (int q = 0; q < DP / V; q++) {
            if ((q + 1) * V > rko0 && q * V < rko1) {
               vec t;
#pragma unroll
               for (int i = 0; i < V; i++) t[i] = (rkg > 0 && q * V + i > rkg) ? Real(0) : v[q * V + i]; // (pad columns beyond the ghost column)
               *(vec *)(pl + q * V) = t;
            }
         }
      } else {
#pragma unroll
         for (int k = 0; k < DP; k++)
            if (k >= rko0 && k < rko1) pl[(int64_t)k * sn] = v[k];
      }
   };
   // the branch state and parameters of a pencil's first frequency-dependent node, ahead of its stage 1
   auto fd_fetch = [&](const uint4 E, Real(&v)[12], Real(&g)[12], Real &sf, Real &u2, Real &x1v, int32_t &k) __attribute__((always_inline)) {
      if (NODES && (E.w & 31u) != 0u) {
         const int32_t li = (int32_t)(E.w >> 8);
#pragma unroll
         for (int m = 0; m < 12; m++)
            if (m < MC) { v[m] = wp.sv_in[st_idx(m, li)]; g[m] = wp.sg_in[st_idx(m, li)]; }
         sf = wp.ssaf[li];
         k = wp.mat[li];
         u2 = wp.x2[li];
         x1v = wp.x1[li];
      }
   };
   // neighbours in file order (+x -x +y -y +z -z) from the pencil (np, nm), march (mp, mm) and lane (lp, lm) axes
   auto air = [&](Real cc, Real old, Real np_, Real nm, Real mp, Real mm, Real lp, Real lm) __attribute__((always_inline)) {
      if (MODE == 2) return upd7<SG>(a1, a2, cc, old, mp, mm, lp, lm, np_, nm);
      if (MODE == 1) return upd7<SG>(a1, a2, cc, old, mp, mm, np_, nm, lp, lm);
      return upd7<SG>(a1, a2, cc, old, np_, nm, mp, mm, lp, lm);
   };
   auto rigid = [&](uint32_t adj, Real cc, Real old, Real np_, Real nm, Real mp, Real mm, Real lp, Real lm) __attribute__((always_inline)) {
      Real nb[6];
      if (MODE == 2) { nb[0] = mp; nb[1] = mm; nb[2] = lp; nb[3] = lm; nb[4] = np_; nb[5] = nm; }
      else if (MODE == 1) { nb[0] = mp; nb[1] = mm; nb[2] = np_; nb[3] = nm; nb[4] = lp; nb[5] = lm; }
      else { nb[0] = np_; nb[1] = nm; nb[2] = mp; nb[3] = mm; nb[4] = lp; nb[5] = lm; }
      return upd_rigid<SG, 6>(a2, wp.sl2, adj, cc, old, nb); // (cpu_engine.h:234-257)
   };

   // One update of the pencil cells 1 .. DP-2 at march coordinate m: Out = f(Cur; Prv, Nxt = the march planes before / after;
   // Old = the value two steps back).  STAGE 1: u^n -> u^{n+1} (halo cells included, their nodes read-only); STAGE 2: owned
   // cells only.  Fv / Fg / Fsf / Fu2 / Fk: branch state and parameters of the pencil's first frequency-dependent node; its new
   // state goes to Fvo / Fgo (may be Fv / Fg), its new value to nval, and st says whether this lane owns it -- NOTHING is stored
   // here (the caller stores after it has consumed the loads in flight: gfx9 counts stores and loads in one in-order counter).
   auto update = [&](auto stage, int m, const Real(&Prv)[DP], const Real(&Cur)[DP], const Real(&Nxt)[DP], const Real(&Old)[DP], const uint4 E,
                     Real(&Out)[DP], bool own_m, const Real(&Fv)[12], const Real(&Fg)[12], Real(&Fvo)[12], Real(&Fgo)[12], Real Fsf, Real Fu2,
                     int32_t Fk, Real &nval, bool &st) __attribute__((always_inline)) {
      constexpr int STAGE = decltype(stage)::value;
      if constexpr (FAST) {
         // the block's structure is uniform and known (usx: node cells, usz: their adjacency, usw: the frequency-dependent one):
         // one scalar bit test per cell decides between the air and the rigid update; the ABC loss can only apply next to the
         // pencil's ghost cell: pencil cell 1 of a low-side region (tested in place), any cell of a high-side one (its ghost sits
         // wherever the aligned pencil puts it: applied after the loop through selects -- one copy of the double-precision
         // division instead of one per pencil cell)
         st = false;
         const uint32_t sx = NODES ? usx : 0u, sw5 = NODES ? (usw & 31u) : 0u, sk0 = usw >> 8;
         Real pfd = Real(0);
#pragma unroll
         for (int k = 1; k < DP - 1; k++) {
            const Real cc = Cur[k];
            const Real lm = lane_from_lower<true>(cc), lp = lane_from_upper<true>(cc);
            // the air update of EVERY cell, straight through; the few node cells (two or three of a pencil) replace theirs out of line:
            // a taken branch per cell and stage -- 72 per march step of a 20-cell pencil -- was a fifth of a wave's time (one wave per SIMD:
            // nothing hides the refill of the instruction buffer)
            Real p = air(cc, Old[k], Cur[k + 1], Cur[k - 1], Nxt[k], Prv[k], lp, lm);
            if (k == 1) {
               const int nk = rnbase + k;
               const bool abc1 = CG ? (!HI && ng_lo) : ((ng_lo && nk == 1) || (ng_hi && nk == NN - 2));
               if (__builtin_expect(abc1, 0)) p = abc_loss<SG>(p, Old[k], wp.l); // (cpu_engine.h:225-229)
            }
            if (NODES && __builtin_expect(((sx >> k) & 1u) != 0u, 0)) {
               const uint32_t jn = __popc(sx & ((1u << k) - 1u));
               p = rigid((usz >> (6 * jn)) & 63u, cc, Old[k], Cur[k + 1], Cur[k - 1], Nxt[k], Prv[k], lp, lm);
               if ((uint32_t)k == sk0) pfd = p;
            }
            Out[k] = p;
         }
         if constexpr (CG) { // (high side: the ABC cell is pencil cell DP - 2, beside the ghost cell; low side: cell 1, above)
            if (HI && ng_hi && !(NODES && ((sx >> (DP - 2)) & 1u))) Out[DP - 2] = abc_loss<SG>(Out[DP - 2], Old[DP - 2], wp.l);
         } else {
            const int kh = NN - 2 - rnbase;
            if (ng_hi && kh >= 2 && kh <= DP - 2 && !(NODES && ((sx >> kh) & 1u))) {
               const Real t = abc_loss<SG>(wall_sel<Real, DP>(Out, kh), wall_sel<Real, DP>(Old, kh), wp.l);
#pragma unroll
               for (int k = 2; k < DP - 1; k++) Out[k] = (k == kh) ? t : Out[k];
            }
         }
         if (NODES && sw5 != 0u) { // (cpu_engine.h:290-301, 363-405) the pencils' frequency-dependent node: state in registers
            const bool owner = own_m && own_lane && (int)sk0 >= rko0 && (int)sk0 < rko1;
            if (eval_lane && (STAGE < NS || STAGE == 1 || owner)) { // (all but the last stage: the halo's nodes too, their state private)
               pfd = fd_regs<Real, MC>(pfd, Fu2, Fsf, Fk, Fv, Fg, Fvo, Fgo, lds, wp.lo2);
               st = owner;
               nval = pfd;
            }
#pragma unroll
            for (int k = 1; k < DP - 1; k++) Out[k] = ((uint32_t)k == sk0) ? pfd : Out[k];
         }
         mirror(Out);
         return;
      }
      const int qm = FAST ? 0 : (mx ? (((wp.first && m == 1) || (wp.last && m == wp.Nx - 2)) ? 1 : 0) : ((m == 1 || m == wp.Ny - 2) ? 1 : 0));
      st = false;
      // Do all the pencils of the wave look alike (same node cells, same adjacency, at most one frequency-dependent node, all
      // within the entry)?  Walls away from edges and corners: then the structure is decoded once, in scalar registers.
      const uint32_t sx = !NODES ? 0u : FAST ? usx : __builtin_amdgcn_readlane(E.x, 1), sz = FAST ? usz : __builtin_amdgcn_readlane(E.z, 1);
      const uint32_t sw5 = !NODES ? 0u : FAST ? (usw & 31u) : (__builtin_amdgcn_readlane(E.w, 1) & 31u), sk0 = FAST ? (usw >> 8) : (__builtin_amdgcn_readlane(E.y, 1) >> 27);
      bool alike = true;
      if (!FAST) {
         const bool differs = eval_lane && (E.x != sx || E.z != sz || (E.w & 31u) != sw5 || (E.y >> 27) != sk0);
         alike = __ballot(differs) == 0ull && __popc(sx) <= 5 && __popc(sw5) <= 1;
      }
      Real pfd = Real(0);
#pragma unroll
      for (int k = 1; k < DP - 1; k++) {
         const Real cc = Cur[k];
         const Real lm = lane_from_lower<true>(cc), lp = lane_from_upper<true>(cc);
         Real p = Real(0);
         const int nk = rnbase + k;
         const bool node_all = alike && ((sx >> k) & 1u);          // a boundary node in every pencil of the wave
         if (!node_all) {
            p = air(cc, Old[k], Cur[k + 1], Cur[k - 1], Nxt[k], Prv[k], lp, lm);
            const int qnm = (((ng_lo && nk == 1) || (ng_hi && nk == NN - 2)) ? 1 : 0) + qm;
            if (qnm > 0 || tile_ql) {
               const int Q = qnm + (FAST ? 0 : ql);
               if (Q > 0) p = abc_loss<SG>(p, Old[k], wp.l * (Real)Q); // (cpu_engine.h:225-229)
            }
         }
         if (node_all) {
            const uint32_t jn = __popc(sx & ((1u << k) - 1u));
            p = rigid((sz >> (6 * jn)) & 63u, cc, Old[k], Cur[k + 1], Cur[k - 1], Nxt[k], Prv[k], lp, lm);
            if (sw5 != 0u && (uint32_t)k == sk0) pfd = p;
         }
         Out[k] = p;
      }
      if (alike) {
         if (sw5 != 0u) { // (cpu_engine.h:290-301, 363-405) the pencils' frequency-dependent node: state in registers
            const bool owner = own_m && own_lane && (int)sk0 >= rko0 && (int)sk0 < rko1;
            if (eval_lane && (STAGE == 1 || owner)) {
               pfd = fd_regs<Real, MC>(pfd, Fu2, Fsf, Fk, Fv, Fg, Fvo, Fgo, lds, wp.lo2);
               st = owner;
               nval = pfd;
            }
#pragma unroll
            for (int k = 1; k < DP - 1; k++)
               if ((uint32_t)k == sk0) Out[k] = pfd;
         }
      } else if (!FAST) {
         // edges, corners, tiles that straddle them: the union of the lanes' node masks, one pencil cell per turn
         uint32_t ub = 0;
#pragma unroll
         for (int k = 1; k < DP - 1; k++)
            if (__ballot((E.x >> k) & 1u) != 0ull) ub |= 1u << k;
         const int k0 = (int)(E.y >> 27);                          // pencil cell of the first frequency-dependent node
         uint32_t jm = 0u;                                         // pencil cells of this lane whose branch ODEs are parked in `jobs`
         constexpr bool BATCH = DP <= 8;                           // (8 job slots per lane)
         while (ub) {
            const int k = __ffs(ub) - 1;
            ub &= ub - 1u;
            const bool has = ((E.x >> k) & 1u) != 0u;
            const int jn = __popc(E.x & ((1u << k) - 1u));         // which node of the pencil
            const bool inl = jn < 5;
            const bool lossy_inl = inl && ((E.w >> jn) & 1u) != 0u;
            const bool prim = has && lossy_inl && k == k0;         // the one whose state is in registers
            const bool need_rec = has && (!inl || (lossy_inl && !prim));
            uint32_t rec = 0u;
            if (__ballot(need_rec) != 0ull) {
               if (need_rec) rec = wp.rec[(E.y & 0x7ffffffu) + (uint32_t)jn];
            }
            const uint32_t adj = inl ? ((E.z >> (6 * jn)) & 63u) : (rec & 63u);
            const Real cc = wall_sel<Real, DP>(Cur, k), nm = wall_sel<Real, DP>(Cur, k - 1), np_ = wall_sel<Real, DP>(Cur, k + 1);
            const Real mp = wall_sel<Real, DP>(Nxt, k), mm = wall_sel<Real, DP>(Prv, k), old = wall_sel<Real, DP>(Old, k);
            const Real lm = lane_from_lower<true>(cc), lp = lane_from_upper<true>(cc);
            Real p = rigid(adj, cc, old, np_, nm, mp, mm, lp, lm);
            const bool owner = own_m && own_lane && k >= rko0 && k < rko1;
            if (prim && (STAGE == 1 || owner)) {
               p = fd_regs<Real, MC>(p, Fu2, Fsf, Fk, Fv, Fg, Fvo, Fgo, lds, wp.lo2);
               st = owner;
               nval = p;
            }
            // any further frequency-dependent node of the pencil: through memory, stored right here
            const bool fds = has && !prim && (inl ? lossy_inl : (rec & 64u) != 0u) && (STAGE == 1 || owner);
            if (BATCH) {
               if (fds) { // parked: the value after the rigid update, the node's place, who stores
                  jobs->p[lane * 8 + k] = p;
                  jobs->li[lane * 8 + k] = (int32_t)(rec >> 8) | (owner ? (1 << 30) : 0);
                  jm |= 1u << k;
               }
            } else if (__ballot(fds) != 0ull) {
               if (fds) {
                  const int32_t li = (int32_t)(rec >> 8);
                  const Real u2 = STAGE == 1 ? wp.x2[li] : wp.x1[li];
                  p = fd_core<Real>(p, u2, li, STAGE == 1 ? wp.sv_in : wp.sv_out, STAGE == 1 ? wp.sg_in : wp.sg_out, wp.sv_out, wp.sg_out, owner, wp.ssaf,
                                    wp.mat, wp.Mb, wp.mq, wp.beta, wp.lo2, wp.mmax);
                  if (owner) (STAGE == 1 ? wp.o1 : wp.o2)[li] = p;
               }
            }
#pragma unroll
            for (int i = 1; i < DP - 1; i++) Out[i] = (has && k == i) ? p : Out[i];
         }
         if (BATCH && __ballot(jm != 0u) != 0ull) {
            jobs->mask[lane] = jm;
            __syncthreads(); // (one wave per block: orders the LDS traffic of its lanes)
            for (int t = 0; t < 8; t++) {
               const int slot = t * 64 + lane, src = slot >> 3, kk = slot & 7;
               const bool valid = ((jobs->mask[src] >> kk) & 1u) != 0u;
               if (__ballot(valid) != 0ull) {
                  if (valid) {
                     const int32_t w = jobs->li[slot], li = w & 0x3fffffff;
                     const bool own = (w >> 30) != 0;
                     const Real u2 = STAGE == 1 ? wp.x2[li] : wp.x1[li];
                     const Real r = fd_core<Real>(jobs->p[slot], u2, li, STAGE == 1 ? wp.sv_in : wp.sv_out, STAGE == 1 ? wp.sg_in : wp.sg_out, wp.sv_out, wp.sg_out, own,
                                                  wp.ssaf, wp.mat, wp.Mb, wp.mq, wp.beta, wp.lo2, wp.mmax);
                     if (own) (STAGE == 1 ? wp.o1 : wp.o2)[li] = r;
                     jobs->p[slot] = r;
                  }
               }
            }
            __syncthreads();
#pragma unroll
            for (int i = 1; i < DP - 1; i++)
               if ((jm >> i) & 1u) Out[i] = jobs->p[lane * 8 + i];
            __syncthreads(); // (before the next stage parks its jobs)
         }
      }
      // ghost cells of the new field: mirror along the pencil, then along the lanes
      mirror(Out);
      if (STAGE == 1 && tile_lg) {
#pragma unroll
         for (int k = 0; k < DP; k++) {
            const Real t = Out[k];
            const Real up = __shfl(t, lane + 2, 64), dn = __shfl(t, lane - 2, 64);
            Out[k] = lg_lo ? up : (lg_hi ? dn : t);
         }
      }
   };

   Real Bm[DP], Bc[DP], Bn[DP], Bq[DP], Ac[DP], Aq[DP], Vm[DP], Vc[DP], Vn[DP], W[DP];
   Real Xm[NS == 3 ? DP : 1], Xc[NS == 3 ? DP : 1], Y[NS == 3 ? DP : 1]; // NS = 3: u^{n+2}(m-3), u^{n+2}(m-2); stage 3's output
   Real F1v[12], F1g[12], Fqv[12], Fqg[12], F2v[12], F2g[12];
   Real F3v[NS == 3 ? 12 : 1], F3g[NS == 3 ? 12 : 1];
   Real F1sf = 0, F1u2 = 0, F1x1 = 0, Fqsf = 0, Fqu2 = 0, Fqx1 = 0, F2sf = 0, F2u2 = 0, F3sf = 0, F3u2 = 0, F2n1 = 0;
   int32_t F1k = 0, Fqk = 0, F2k = 0, F3k = 0;
#pragma unroll
   for (int q = 0; q < 12; q++) { F1v[q] = F1g[q] = Fqv[q] = Fqg[q] = F2v[q] = F2g[q] = Real(0); if (NS == 3) F3v[q] = F3g[q] = Real(0); }
   // first and last march step: NS stages need u^{n+1} NS - 1 planes before and after the owned ones (NS = 3: u^{n+2} one plane)
   const int mf = NS == 1 ? ms : ms - (NS - 1), ml = NS == 1 ? me - 1 : me + (NS - 2);
   uint4 Epp = make_uint4(0u, 0u, 0u, 0u), Ep = make_uint4(0u, 0u, 0u, 0u), Ec = mask_ent(load_ent(mf), mf), En = mask_ent(load_ent(mf + 1), mf + 1), Eq;
   load_pencil(wp.B, mf - 1, Bm);
   load_pencil(wp.B, mf, Bc);
   load_pencil(wp.B, mf + 1, Bn);
   load_pencil(wp.A, mf, Ac);
   mirror(Bm); mirror(Bc); mirror(Bn);
   fd_fetch(Ec, F1v, F1g, F1sf, F1u2, F1x1, F1k);
#pragma unroll
   for (int k = 0; k < DP; k++) { Vm[k] = Real(0); Vc[k] = Real(0); Vn[k] = Real(0); W[k] = Real(0); }
   if constexpr (NS == 3) {
#pragma unroll
      for (int k = 0; k < DP; k++) { Xm[k] = Real(0); Xc[k] = Real(0); Y[k] = Real(0); }
   }
   // before the loop: what march step mf needs next
   Eq = load_ent(mf + 2);
   load_pencil(wp.B, mf + 2, Bq);
   load_pencil(wp.A, mf + 1, Aq);
   fd_fetch(En, Fqv, Fqg, Fqsf, Fqu2, Fqx1, Fqk);
   for (int m = mf; m <= ml; m++) {
      usx = dsx; usz = dsz; usw = dsw;
      if constexpr (CG) asm volatile("" : "+s"(usx), "+s"(usz), "+s"(usw));
      else {
         rkg = R.kg; rko0 = R.ko0; rko1 = R.ko1; rkb0 = R.kb0; rkb1 = R.kb1; rnbase = R.nbase;
         asm volatile("" : "+s"(rkg), "+s"(rko0), "+s"(rko1), "+s"(rkb0), "+s"(rkb1), "+s"(rnbase), "+s"(usx), "+s"(usz), "+s"(usw));
      }
      // stage 1: u^{n+1}(m)
      const bool own_m = m >= ms && m < me;
      Real nv1 = Real(0), nv2 = Real(0), nv3 = Real(0);
      bool st1 = false, st2 = false, st3 = false;
      update(std::integral_constant<int, 1>(), m, Bm, Bc, Bn, Ac, Ec, Vn, own_m, F1v, F1g, F1v, F1g, F1sf, F1u2, F1k, nv1, st1);
      // stage 2: u^{n+2}(m-1) from u^{n+1}(m-2 .. m); a ghost plane of the march axis is the plane two further in
      // (opaque again: otherwise every per-cell predicate of stage 1 is kept for stage 2 -- in vector-register lanes, two
      // v_writelane per cell -- instead of being tested again with one scalar instruction)
      if constexpr (CG) asm volatile("" : "+s"(usx), "+s"(usz), "+s"(usw));
      else asm volatile("" : "+s"(rkg), "+s"(rko0), "+s"(rko1), "+s"(rnbase), "+s"(usx), "+s"(usz), "+s"(usw));
      const bool do2 = NS >= 2 && m - 1 >= ms - (NS - 2);
      const bool own_m2 = m - 1 >= ms && m - 1 < me;
      if (do2) {
         if (!mx && !FAST) {
            const bool sub_hi = m == NM - 1 && mg_hi, sub_lo = m - 2 == 0 && mg_lo;
            Real Pv[DP], Nv[DP];
#pragma unroll
            for (int k = 0; k < DP; k++) { Pv[k] = sub_lo ? Vn[k] : Vm[k]; Nv[k] = sub_hi ? Vm[k] : Vn[k]; }
            update(std::integral_constant<int, 2>(), m - 1, Pv, Vc, Nv, Bm, Ep, W, true, F2v, F2g, F2v, F2g, F2sf, F2u2, F2k, nv2, st2);
         } else update(std::integral_constant<int, 2>(), m - 1, Vm, Vc, Vn, Bm, Ep, W, own_m2, F2v, F2g, F2v, F2g, F2sf, F2u2, F2k, nv2, st2);
      }
      // stage 3 (NS = 3): u^{n+3}(m-2) from u^{n+2}(m-3 .. m-1) -- the last of them stage 2's output just now --, old value u^{n+1}(m-2)
      const bool do3 = NS == 3 && m - 2 >= ms;
      if constexpr (NS == 3) {
         if constexpr (CG) asm volatile("" : "+s"(usx), "+s"(usz), "+s"(usw));
         else asm volatile("" : "+s"(rkg), "+s"(rko0), "+s"(rko1), "+s"(rnbase), "+s"(usx), "+s"(usz), "+s"(usw));
         if (do3) update(std::integral_constant<int, 3>(), m - 2, Xm, Xc, W, Vm, Epp, Y, true, F3v, F3g, F3v, F3g, F3sf, F3u2, F3k, nv3, st3);
      }
      const int32_t li1 = (int32_t)(Ec.w >> 8), li2 = (int32_t)(Ep.w >> 8), li3 = (int32_t)(Epp.w >> 8);
      // Rotate.  What was loaded a march step ago is touched HERE, before anything is stored: gfx9 counts loads and stores in one
      // in-order counter, so a wait for those loads placed after this step's stores would wait for the stores as well (and
      // without the touch the copies below are mere renamings: the first real use, and the wait, would be in the next step).
#pragma unroll
      for (int k = 0; k < DP; k++) { Vm[k] = Vc[k]; Vc[k] = Vn[k]; Bm[k] = Bc[k]; Bc[k] = Bn[k]; Bn[k] = Bq[k]; Ac[k] = Aq[k]; }
#pragma unroll
      for (int k = 0; k < DP; k++) { asm volatile("" : "+v"(Bn[k])); asm volatile("" : "+v"(Ac[k])); }
#pragma unroll
      for (int q = 0; q < 12; q++) { asm volatile("" : "+v"(Fqv[q])); asm volatile("" : "+v"(Fqg[q])); }
      asm volatile("" : "+v"(Fqsf), "+v"(Fqu2), "+v"(Fqx1), "+v"(Fqk), "+v"(Eq.x), "+v"(Eq.y), "+v"(Eq.z), "+v"(Eq.w) : : "memory");
      mirror(Bn);
      // the stores of this march step ...
      if (own_m) store_pencil(wp.C, m, Vc);
      if (do2 && own_m2) store_pencil(wp.D, m - 1, W);
      if constexpr (NS == 3) { if (do3) store_pencil(wp.E, m - 2, Y); }
      if (st1) wp.o1[li1] = nv1;
      if (NS == 1 && st1) { // one step: the state after stage 1 is the state
#pragma unroll
         for (int q = 0; q < 12; q++)
            if (q < MC) { wp.sv_out[st_idx(q, li1)] = F1v[q]; wp.sg_out[st_idx(q, li1)] = F1g[q]; }
      }
      if (st2) {
         wp.o2[li2] = nv2;
         if (NS == 2) {
#pragma unroll
            for (int q = 0; q < 12; q++)
               if (q < MC) { wp.sv_out[st_idx(q, li2)] = F2v[q]; wp.sg_out[st_idx(q, li2)] = F2g[q]; }
         }
      }
      if constexpr (NS == 3) {
         if (st3) {
            wp.o3[li3] = nv3;
#pragma unroll
            for (int q = 0; q < 12; q++)
               if (q < MC) { wp.sv_out[st_idx(q, li3)] = F3v[q]; wp.sg_out[st_idx(q, li3)] = F3g[q]; }
         }
      }
      asm volatile("" : : : "memory");
      if constexpr (NS == 3) {
#pragma unroll
         for (int k = 0; k < DP; k++) { Xm[k] = Xc[k]; Xc[k] = W[k]; }
#pragma unroll
         for (int q = 0; q < 12; q++) { F3v[q] = F2v[q]; F3g[q] = F2g[q]; }
         F3sf = F2sf; F3k = F2k; F3u2 = F2n1; // (stage 3's u2b: the node's own u^{n+1}, from its stage 1 two march steps ago)
         F2n1 = nv1;
      }
#pragma unroll
      for (int q = 0; q < 12; q++) { F2v[q] = F1v[q]; F2g[q] = F1g[q]; F1v[q] = Fqv[q]; F1g[q] = Fqg[q]; }
      F2sf = F1sf; F2u2 = F1x1; F2k = F1k;
      F1sf = Fqsf; F1u2 = Fqu2; F1x1 = Fqx1; F1k = Fqk;
      Epp = Ep; Ep = Ec; Ec = En; En = mask_ent(Eq, m + 2);
      // ... and the loads of the one after the next
      if (m + 1 <= ml) {
         Eq = load_ent(m + 3);
         if (m + 1 < ml) {
            load_pencil(wp.B, m + 3, Bq);
            load_pencil(wp.A, m + 2, Aq);
            fd_fetch(En, Fqv, Fqg, Fqsf, Fqu2, Fqx1, Fqk);
         }
      }
   }
}

// One launch = a list of blocks (WallParams::blk: region | lane tile << 3 | march chunk << 16, and for FAST launches the common
// structure of the block's pencils).  !VEC: regions normal to x and y (lanes along z); VEC: regions normal to z.
// MC = how many branch states per node the kernel moves and evaluates, a compile-time bound of the scene's largest branch count
// (4 or 12; the state arrays hold 12 slots per node, pf_kernels.h: st_idx): with the count itself, a kernel argument, as the
// bound every branch m sat in a block of its own -- compare, jump, reload of the array pointers, wait -- 12 times per fetch, per
// evaluation and per store.  Slots between the scene's count and MC are loaded and stored back unchanged.
// SG: the reference GPU engine's safeguarded arithmetic (pf_kernels.h: upd7 / upd_rigid / abc_loss<true>) instead of the C CPU engine's.
template <typename Real, int DP, bool VEC, bool FAST, bool NODES = true, int MC = 12, bool SG = false, int NS = 2, int GD = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((FAST && !VEC && NS != 3 && sizeof(Real) == 4) ? 2 : 1))) void k_wall2(WallParams<Real> wp, Real a1, Real a2) {
   static_assert(FAST || NODES, "generic blocks have everything");
   const uint4 bd = wp.blk[blockIdx.x];
   const WallRegion R = wp.reg[bd.x & 7u];
   const int j = (int)((bd.x >> 3) & 0x1fffu), c = (int)(bd.x >> 16);
   __shared__ WallLds<Real> lds;
   __shared__ typename std::conditional<FAST, int, WallJobs<Real>>::type jobs_mem; // (generic blocks only)
   WallJobs<Real> *jobs = nullptr;
   if constexpr (!FAST) {
      jobs = &jobs_mem;
      jobs->mask[threadIdx.x] = 0u;
   }
   if (NODES) {
      for (int i = threadIdx.x; i < wp.nmat * 12; i += 64) lds.mq[i] = wp.mq[i];
      for (int i = threadIdx.x; i < wp.nmat; i += 64) { lds.beta[i] = wp.beta[i]; lds.M[i] = wp.Mb[i]; }
      __syncthreads();
   }
   if constexpr (GD > 0) { // constant pencil geometry: a low-side and a high-side body
      const bool hi = R.kg != 0;
      if constexpr (VEC) { if (hi) wall_body<Real, DP, 2, FAST, NODES, MC, SG, NS, GD, true>(wp, R, j, c, a1, a2, &lds, bd.y, bd.z, bd.w, jobs);
                           else wall_body<Real, DP, 2, FAST, NODES, MC, SG, NS, GD, false>(wp, R, j, c, a1, a2, &lds, bd.y, bd.z, bd.w, jobs); }
      else if (R.mode == 1) { if (hi) wall_body<Real, DP, 1, FAST, NODES, MC, SG, NS, GD, true>(wp, R, j, c, a1, a2, &lds, bd.y, bd.z, bd.w, jobs);
                              else wall_body<Real, DP, 1, FAST, NODES, MC, SG, NS, GD, false>(wp, R, j, c, a1, a2, &lds, bd.y, bd.z, bd.w, jobs); }
      else { if (hi) wall_body<Real, DP, 0, FAST, NODES, MC, SG, NS, GD, true>(wp, R, j, c, a1, a2, &lds, bd.y, bd.z, bd.w, jobs);
             else wall_body<Real, DP, 0, FAST, NODES, MC, SG, NS, GD, false>(wp, R, j, c, a1, a2, &lds, bd.y, bd.z, bd.w, jobs); }
   } else if constexpr (VEC) wall_body<Real, DP, 2, FAST, NODES, MC, SG, NS>(wp, R, j, c, a1, a2, &lds, bd.y, bd.z, bd.w, jobs);
   else if (R.mode == 1) wall_body<Real, DP, 1, FAST, NODES, MC, SG, NS>(wp, R, j, c, a1, a2, &lds, bd.y, bd.z, bd.w, jobs);
   else wall_body<Real, DP, 0, FAST, NODES, MC, SG, NS>(wp, R, j, c, a1, a2, &lds, bd.y, bd.z, bd.w, jobs);
}

} // namespace pf
