// pf_engine.hip -- host side of libpffdtd_hip.so: the C ABI of include/pffdtd_hip.h over the kernels of
// pf_kernels.h.  Replaces the reference's `double run_sim(struct SimData*)` (c_cuda/gpu_engine.h:665-1255,
// c_cuda/cpu_engine.h:52-360) for MI355X.  Not derived from gpu_engine.h: different memory layout (padded
// pitch, engine-built skip-mask), different kernels (2.5D register marching), device-resident source signals
// and receiver ring instead of per-step host traffic, and a split-phase step for overlapped slab exchange.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "pffdtd_hip.h"
#include "pf_debug.h"
#include "pf_kernels.h"
#include "pf_air_fused.h"
#include "pf_energy.h"
#include "pf_tb2.h"
#include "pf_tb3.h"
#include "pf_wall.h"
#include "pf_brick.h"

namespace {

thread_local std::string g_err;

int set_err(int code, const char *fmt, ...) {
   char buf[1024];
   va_list ap;
   va_start(ap, fmt);
   vsnprintf(buf, sizeof buf, fmt, ap);
   va_end(ap);
   g_err = buf;
   return code;
}

} // namespace

// internal: lets the other translation units of this library (pf_vox.hip) feed pf_last_error()
extern "C" void pf__set_error(const char *msg) { g_err = msg ? msg : ""; }

// internal (pf_engine.hip, pf_multi.hip): would this scene rather be stored with the file's x and z axes exchanged (Engine::swz)?
// Rooms only -- a box-shaped room (most boundary nodes within a few cells of a grid face) steps in blocked pairs, which exist
// for the file's axis order alone --, when clearly more boundary nodes have their successor ALONG FILE X in the list than
// along file z: those runs become unit-stride runs, the others a row apart.  counts[0..1] = the two run counts.
extern "C" int pf__axis_exchange_pays(const pf_simdata *sd, int64_t *counts) {
   if (counts) counts[0] = counts[1] = 0;
   const int64_t fNx = sd->Nx, fNy = sd->Ny, fNz = sd->Nz;
   if (sd->Nb < 100000 || sd->Npts > ((int64_t)1 << 34) || fNx <= fNz) return 0; // (small scenes: nothing to gain; the bitmap below is Npts / 8 bytes)
   const int64_t NzNy = fNz * fNy;
   std::vector<uint64_t> bits;
   try { bits.assign((size_t)(sd->Npts >> 6) + 1, 0); } catch (const std::bad_alloc &) { return 0; } // (no room for the bitmap: file order; nothing may cross the C ABI)
   int64_t near_face = 0;
   for (int64_t i = 0; i < sd->Nb; i++) {
      const int64_t ii = sd->bn_ixyz[i];
      if (ii < 0 || ii >= sd->Npts) return 0; // (reported by the engine's own checks)
      bits[(size_t)(ii >> 6)] |= (uint64_t)1 << (ii & 63);
      const int64_t fz = ii % fNz, fy = (ii / fNz) % fNy, fx = ii / NzNy;
      const int64_t d = std::min(std::min(std::min(fx, fNx - 1 - fx), std::min(fy, fNy - 1 - fy)), std::min(fz, fNz - 1 - fz));
      near_face += d < 16;
   }
   if (near_face * 10 >= sd->Nb * 8) return 0; // a box-shaped room
   auto has = [&](int64_t ii) { return ii < sd->Npts && ((bits[(size_t)(ii >> 6)] >> (ii & 63)) & 1u) != 0; };
   int64_t run_z = 0, run_x = 0;
   for (int64_t i = 0; i < sd->Nb; i++) {
      const int64_t ii = sd->bn_ixyz[i];
      run_z += has(ii + 1);
      run_x += has(ii + NzNy);
   }
   if (counts) { counts[0] = run_x; counts[1] = run_z; }
   // measured (profiles/r03_reference_configs.jsonl): CTK church 3.46 M / 2.75 M (x / z successors) +12-15 % exchanged, Musikverein
   // 17.2 M / 14.4 M +11 %
   return (double)run_x > 1.1 * (double)run_z && run_x - run_z > sd->Nb / 20;
}

namespace {

#define HIPCHK(expr)                                                                                          \
   do {                                                                                                       \
      hipError_t _e = (expr);                                                                                 \
      if (_e != hipSuccess)                                                                                   \
         return set_err(PF_ERR_HIP, "HIP error %s at %s:%d: %s", hipGetErrorName(_e), __FILE__, __LINE__,     \
                        hipGetErrorString(_e));                                                               \
   } while (0)

inline int64_t round_up(int64_t a, int64_t m) { return (a + m - 1) / m * m; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// z pitch: rows start on 128-byte lines.  (Measured and dropped: one extra line on pitches that are a multiple of 4 KiB,
// to spread a column's rows over more memory channels -- every kernel got slower, lean 2.31 -> 2.53 ms, 13-point 2.45 -> 2.58.
// Re-measured in round 3 on the pair path at 1024^3, alternating runs: 472-476 Gvox/s without, 455 with one extra line, 435 with
// two; per kernel (rocprofv3, pad 0 -> 1): pair kernel 3.06 -> 3.03 ms, boundary launch 0.411 -> 0.384 (its scattered stores do
// spread over more channels), but the column-strip kernel 0.243 -> 0.343 (its right strip then spans the pad columns too) and the
// lean kernel pays a fifth, mostly empty segment: a net loss unless those two learn about the pad, worth 2 % at best.  Nor does
// a padded pitch remove the grid-placement lottery: 40 candidates of the search span 2.939 ... 3.42 ms per launch with it
// (median 3.15) against 2.975 ... 3.65 without (median 3.23).)
int64_t grid_pitch(int64_t Nz, int32_t real_bytes) { return round_up(Nz, 128 / real_bytes); }

// DPP wave-shift semantics verified once per process on the device
int dpp_ok_cached = -1;
int check_dpp(hipStream_t s) {
   if (dpp_ok_cached >= 0) return dpp_ok_cached;
   int *d = nullptr, h = 0;
   if (hipMalloc(&d, sizeof(int)) != hipSuccess) return 0;
   hipLaunchKernelGGL(pf::k_dpp_selftest, dim3(1), dim3(64), 0, s, d);
   hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, s);
   hipStreamSynchronize(s);
   hipFree(d);
   dpp_ok_cached = h;
   return h;
}

struct Range { int64_t b, e; };

// The creation-time measurements (kernel choice, grid placement) of engines that share a device run one at a time: one engine's
// temporary candidate grids (up to 85 % of the device, pool_extra) must not starve another's mandatory allocations, and
// measurements taken side by side would time each other's kernels.
std::mutex g_tune_mu[64];

struct EngineBase {
   virtual ~EngineBase() {}
   virtual int run(int64_t n0, int64_t nsteps) = 0;
   virtual int step_begin(int64_t n) = 0;
   virtual int step_end(int64_t n) = 0;
   virtual int halo_ptrs(void **slo, void **shi, void **rlo, void **rhi, size_t *bytes) = 0;
   virtual int state_grids(void **up, void **uc) = 0;
   virtual int layout(int64_t *dims, int64_t *pitch, int32_t *exchanged) = 0;
   virtual int sync() = 0;
   virtual int flush() = 0;
   virtual int set_spares(void *g2, void *g3) = 0;
   virtual int place_grids(void *const *grids, int n, int32_t *idx) = 0;
   virtual int place_grids5(void *const *grids, int n, int32_t *idx) = 0;
   virtual int get_grid(int which, void *host) = 0;
   virtual int set_grid(int which, const void *host) = 0;
   virtual int timing(pf_timing *t, int reset) = 0;
   virtual int set_timing(int on) = 0;
   virtual void *stream(int which) = 0;
   virtual int energy_cfg(double h, double c, double Ts, const double *DEF) = 0;
   virtual int run_energy(int64_t n0, int64_t nsteps, double *H, double *El, double *Ei) = 0;
};

template <typename Real> struct Engine : EngineBase {
   pf_simdata sd{};
   pf_opts op{};
   int64_t Nx = 0, Ny = 0, Nz = 0, P = 0, plane = 0, npad = 0; // STORAGE dimensions (= the file's unless swz)
   // Axis exchange: the reference's GPU preparation sorts the axes by size (rotate_sim_data.py:30-130), which makes the SMALLEST
   // dimension the unit-stride one -- and the room's largest surfaces (floor, ceiling: normal to it) the ones whose nodes lie a
   // whole row apart, a 128-byte line of u^n, u^{n-1} and u^{n+1} per node in the boundary pass.  With swz the engine STORES
   // the grid with the file's x and z axes exchanged (unit stride along file x, the longest axis): the strided surfaces are
   // then the smallest ones.  Kernels work in storage coordinates; the order in which neighbours enter the sums, the adjacency
   // bits and every index the caller sees stay in file terms, so the bits do not change.  Single-domain engines with their
   // own grids only (a slab's ghost planes must be contiguous; caller-owned grids have the documented layout).
   bool swz = false;
   int64_t fNx = 0, fNy = 0, fNz = 0;                          // the file's dimensions
   int64_t Nb = 0, Nbl = 0, Nba = 0, Ns = 0, Nr = 0, Nt = 0;
   int mb_max = 0; // largest branch count of the materials
   bool fcc = false, fold = false;
   bool use_dpp = true;
   Real a1, a2, sl2, lo2, l;
   // device state
   Real *u0 = nullptr, *u1 = nullptr;
   bool own_grids = true;
   std::vector<float> place_ms;                           // sample_placement: ms per launch of every candidate
   bool tb2_probe = false;                                // launch_tb2 under its creation-time name (k_tb2_reg<..., PROBE>)
   std::vector<Real *> own_list;
   uint8_t *mask = nullptr;      // skip-mask (boundary nodes + ghost z + pad + parity)
   Real *v1_dst = nullptr;       // autotune: destination of the barrier-free 7-point kernel (null = in place)
   int lw_force = 0;             // autotune: lanes per row segment of the barrier-free kernels (0 = pick_lw's rule)
   int order_force = -1;         // autotune: tile order of the marching kernels (-1 = swizzle_mode's rule; 0 plain, 2 XCD-banded)
   float tune_ms[3] = {0, 0, 0}; // measured at creation: lean / barrier-free / blocked pair (per step), ms
   float pair_margin = 0.99f;    // the pair path stays when it takes less than this fraction of the best single step
   bool lean = false, need_fold_row = false; // lean: the fused 7-point kernel of pf_air_fused.h (air_variant 25)
   bool vg = false;          // barrier-free marching kernel with virtual ghost shell + in-kernel ABC (air_variant 4)
   bool abck = false;        // barrier-free marching kernel with memory flips but the ABC loss in-kernel (air_variant 7)
   int vbase = 0;            // air_variant without its flag bit (256: separate rigid / branch-ODE kernels)
   bool sg = false;          // PF_NUM_GPU_SAFEGUARDED
   int lean_nzt = 0;
   int64_t *d_bn = nullptr, *d_bnl = nullptr, *d_bna = nullptr, *d_in = nullptr, *d_out = nullptr;
   uint16_t *d_adj = nullptr;
   int32_t *d_lossy = nullptr;   // per boundary node: index into the lossy-node arrays or -1 (fused boundary pass)
   bool fuse_boundary = false;
   int8_t *d_Q = nullptr, *d_mat = nullptr, *d_Mb = nullptr;
   Real *d_ssaf = nullptr, *d_beta = nullptr, *d_insig = nullptr;
   pf::MatQuadT<Real> *d_mq = nullptr;
   Real *ub[3] = {nullptr, nullptr, nullptr}; // u0b, u1b, u2b (cpu_engine.h:94-96), rotated each step
   Real *u2ba = nullptr, *vh1 = nullptr, *gh1 = nullptr;
   Real *ring = nullptr;
   Real *h_ring = nullptr; // pinned
   int64_t ring_depth = 0, ring_fill = 0, ring_n0 = 0;
   std::vector<int64_t> out_row; // sorted receiver slot -> caller row
   // plane ranges of the sorted lists: lo = first owned plane (ix==1), hi = last owned plane (ix==Nx-2)
   Range bn_lo, bn_mid, bn_hi, bnl_lo, bnl_mid, bnl_hi, bna_lo, bna_mid, bna_hi, in_lo, in_mid, in_hi;
   // the same lists cut for the split-phase pairs, whose edge stream owns two planes per side: planes 1-2 / 3..Nx-4 / Nx-3..Nx-2
   Range bn_lo2, bn_mid2, bn_hi2, bnl_lo2, bnl_mid2, bnl_hi2, in_lo2, in_mid2, in_hi2;
   // ... and for the split-phase triples, three planes per side: planes 1-3 / 4..Nx-5 / Nx-4..Nx-2
   Range bn_lo3, bn_mid3, bn_hi3, bnl_lo3, bnl_mid3, bnl_hi3, in_lo3, in_mid3, in_hi3;
   hipStream_t s_main = nullptr, s_edge = nullptr, s_wall = nullptr, s_wall2 = nullptr; // s_wall, s_wall2: a slab's wall regions, alike / generic blocks (created on first use)
   hipEvent_t ev_pre = nullptr, ev_edge = nullptr, ev_main = nullptr, ev_wall0 = nullptr, ev_wall = nullptr, ev_wall2 = nullptr;
   bool wall_pending = false;
   bool in_step = false;
   bool state_touched = false; // a caller wrote the field (pf_engine_set_grid): the placement search, which steps and then zeroes the offered grids, is refused
   int64_t steps_done = 0;
   // launch-bound grids: six steps (the period of the u0/u1 swap and the three-deep u0b ring) captured once in a hipGraph
   // and replayed; the step index and the ring column are read from device counters
   bool graph_ok = false;
   hipGraphExec_t gexec = nullptr;
   int64_t rot_count = 0, g_rot0 = -1;
   int64_t *d_ctr = nullptr;
   // temporal blocking (pf_tb2.h): pairs of steps over a boundary-free box, single-step strips around it
   bool tb2 = false;                                      // pairs inside pf_engine_run (single-domain engines)
   bool tb2_geom = false, tb2_slab = false;               // slab engines: pairs across two split-phase steps (set_spares)
   int pair_phase = 0;                                    // 1: between the two steps of a split-phase pair
   bool pair_now = false;                                 // the step in flight is half of a pair
   bool triple_now = false;                               // ... a third of a triple (tb3_slab; pair_phase then counts 0, 1, 2)
   Real *pA = nullptr, *pB = nullptr;                     // u^{n-1}, u^n of the pair in flight
   Real *bufC = nullptr, *bufD = nullptr;                 // the two extra state grids of the out-of-place pair
   // three steps per pass (pf_tb3.h, Engine::step_triple): single-domain 7-point engines whose shell steps as wall regions.  Five
   // grids: the state (u^{n-1}, u^n) -> bufD = u^{n+2}, bufE = u^{n+3}; bufC holds u^{n+1} where somebody needs it in memory (the
   // shell, the single-step tiles and their neighbours)
   bool tb3 = false;
   bool tb3_geom = false, tb3_slab = false;               // slab engines: the box and its tiles are k_tb3's / triples across three split-phase steps (place_grids5)
   bool triples() const { return tb3 || tb3_slab; }
   Real *bufE = nullptr;
   Real *home[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // the placed role cycle: state (0, 1) <-> targets (2, 3), 4 = the u^{n+1} grid
   static constexpr int tb3_wt = 8, tb3_r = 3, tb3_rows = tb3_wt * tb3_r - 4; // k_tb3<Real, 3, 8>: 20 core rows per tile
   int tbx0 = 0, tbx1 = 0, tby0 = 0, tby1 = 0, tbz0 = 0, tbz1 = 0; // box of cells k_tb2_reg produces
   int tbzu0 = 0, tbzu1 = 0;                                        // ... its z range before it was moved to whole vectors / tiles (the frame's bricks end here)
   int szl = 0, szr = 0;                                  // 7-point column strips: columns [0, szl) and [szr, P)
   // planes per x chunk of k_tb2_reg: 12-20 are equally fast, 24 is 1 % and 48 is 6 % slower although longer chunks
   // re-read fewer prologue planes (1024^3, tools/tb2_probe.py)
   int tb2_chunk = 16;
   std::vector<std::pair<int, int>> tb_xr;                // its x range (empty: no box)
   // the box is cut into tiles (x chunk x rows of one workgroup x core columns of one row segment); tiles with a boundary
   // node or a source within one cell of their core ("dirty") take single steps (k_tb1_tile), the others pairs
   int tb_lw = 64, tb_chunk = 16, tb_nxc = 0, tb_nyt = 0, tb_nzt = 0;
   int32_t *tb_clean = nullptr, *tb_dirty = nullptr;      // tile ids (xc*nyt + yt)*nzt + zt
   int32_t *tb_sample = nullptr;                          // placement search: the clean tiles of every k-th x chunk (same order)
   int64_t tb_nsample = 0;
   double tb_sample_frac = 1.0;                           // their share of the clean cells
   int64_t tb_nclean = 0, tb_ndirty = 0, tb_clean_cells = 0;
   bool tb_order_band = false;
   // 13-point pairs (folded FCC): whatever of the box is not a clean tile's core is stepped by k_air_fcc over its own tiles
   // (256 columns x 16 rows x the same x chunks), listed here
   int32_t *sh_tiles = nullptr;
   static constexpr int fcc_wt = 8; // waves per workgroup of k_tb2_fcc_x (two of them halo providers)
   int64_t sh_ntiles = 0;
   int sh_nyt = 0, sh_nzt = 0;
   const Real *u0_src = nullptr;                          // out-of-place single-step launches read u^{n-1} here
   int lean_yt0 = 0, lean_nyt = -1;                       // row-strip launches of the lean kernel (-1: all tiles)
   int lean_x2_begin = 0, lean_x2_end = 0;                // a second x slab for the next lean launch (launch_shell_rest)
   // boundary nodes inside the column strips are updated by k_air_zstrip itself (it streams their lines anyway; in
   // the list kernel the floor / ceiling nodes of a box room cost half of the whole boundary pass)
   uint32_t *zs_map = nullptr;                            // per strip vector: first node number << 4 | node bits (ZStripParams::zvec)
   uint16_t *zs_adj = nullptr;                            // adjacency bits / lossy-list positions of the strips' nodes, in strip order
   int32_t *zs_li = nullptr;
   int32_t *zs_rest = nullptr;                            // the other boundary nodes (positions in the boundary list)
   int64_t zs_nrest = 0;
   int zs_mode = 0;                                       // 0: the list kernel does them (debug 0x20000000, and the fallback);
                                                          // 2: strip kernel does the rigid update, extra threads of the k_boundary launch the branch ODEs (default)
   int32_t *zs_fd = nullptr;                              // mode 2: the lossy nodes (indices into the lossy arrays) inside the strips
   int64_t zs_nfd = 0;
   const int32_t *bnd_sel = nullptr;                      // launch_boundary visits bnd_sel[range] when set
   // wall regions (pf_wall.h): the shell of a blocked pair -- wall layers, ABC cells, ghost mirrors -- stepped in pairs too
   bool wl_on = false;
   Real *wsP[3] = {nullptr, nullptr, nullptr};            // slab pairs with wall regions: the node-value buffers u0b / u1b / u2b at the start of the pair
   // launch groups: 0 = regions normal to x / y (lanes along z, pencils of 8 cells); 1 / 2 / 3 = regions normal to z (lanes along
   // y) with vector pencils of 12 / 16 / 20 cells.  Each has a list of alike blocks and one of generic blocks.
   struct WlGroup { int nreg = 0; pf::WallRegion reg[pf::WALL_MAXREG]; uint32_t blk0[3] = {0, 0, 0}, nblk[3] = {0, 0, 0}; }; // lists: alike, generic, alike without nodes
   WlGroup wl_grp[4];
   uint4 *wl_blk = nullptr;                               // block lists of the four launches: strided / vector pencils x alike (fast) / generic
   uint4 *wl_pen = nullptr;                               // per pencil: node mask, first record, adjacency / flags of the first five nodes (pf_wall.h)
   uint32_t *wl_rec = nullptr;                            // per node of a pencil: adjacency bits | lossy flag | lossy position
   int32_t *wl_rest = nullptr;                            // boundary nodes no wall region owns (inside the box): the list kernel's
   int64_t wl_nrest = 0;
   // the frame (pf_brick.h): the edges and corners of the shell as bricks stepped in LDS -- instead of the regions' generic blocks
   pf::Brick *wl_brk = nullptr;
   uint32_t *wl_binfo = nullptr;                          // per cell of a brick's extended box: adjacency | node flags | ABC count
   uint2 *wl_blos = nullptr;                              // per frequency-dependent node of a brick: cell | owned << 31, place in the lossy arrays
   int64_t wl_nbrk = 0, wl_nbown_dbg = 0;
   bool wl_ns3 = false, wl_ns3z = false, wl_no_ns3 = false;                // the x / y regions take three steps per pass (k_wall2<..., NS = 3>); ... found impossible for this scene
   Real *ubx[2] = {nullptr, nullptr};                     // single domains with wall regions: two more node-value buffers beside ub[0..2]
   int wl_geo[4] = {0, 0, 0, 0};                          // per launch group: the box margin all its regions' pencils share (standard geometry, pf_wall.h GD), else 0
   int wl_chunk_want[2] = {0, 0};                         // march steps per block the x / y regions' and the column strips' launches aim for (init_walls)
   size_t wl_brk_lds = 0;                                 // dynamic LDS of a brick launch (the largest brick)
   Real *vh1b = nullptr, *gh1b = nullptr;                 // the other half of the double-buffered branch state
   Real *bs_vout = nullptr, *bs_gout = nullptr;           // launch_boundary: where the new branch state goes (null: in place)
   // energy diagnostic (pf_energy.h)
   Real *Lu = nullptr, *vh_old = nullptr, *u2in = nullptr;
   double *d_acc = nullptr, *d_DEF = nullptr;
   double en_h = 0, en_c = 0, en_Ts = 0;
   bool en_ready = false;
   // timing
   std::vector<std::pair<hipEvent_t, hipEvent_t>> air_ev, step_ev, tb2_ev, ev_pool;
   pf_timing tm{};

   ~Engine() override { destroy(); }

   void destroy() {
      if (s_main) hipStreamSynchronize(s_main);
      if (s_edge) hipStreamSynchronize(s_edge);
      auto F = [](void *p) { if (p) hipFree(p); };
      for (Real *g : own_list) F(g); // state grids this engine allocated (u0/u1 unless external, the temporal-blocking spares)
      own_list.clear();
      F(wl_pen); F(wl_rec); F(wl_rest); F(wl_blk); F(wl_brk); F(wl_binfo); F(wl_blos); F(ubx[0]); F(ubx[1]); F(vh1b); F(gh1b); F(d_lossy); F(mask); F(zs_map); F(zs_adj); F(zs_li); F(zs_rest); F(zs_fd); F(tb_clean); F(tb_dirty); F(tb_sample); F(sh_tiles); F(Lu); F(vh_old); F(u2in); F(d_acc); F(d_DEF); F(d_bn); F(d_bnl); F(d_bna); F(d_in); F(d_out); F(d_adj); F(d_Q); F(d_mat); F(d_Mb); F(d_ssaf);
      F(d_beta); F(d_insig); F(d_mq); F(ub[0]); F(ub[1]); F(ub[2]); F(u2ba); F(vh1); F(gh1); F(ring);
      if (h_ring) hipHostFree(h_ring);
      for (auto &p : air_ev) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
      for (auto &p : step_ev) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
      for (auto &p : tb2_ev) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
      for (auto &p : ev_pool) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
      if (ev_pre) hipEventDestroy(ev_pre);
      if (ev_edge) hipEventDestroy(ev_edge);
      if (ev_main) hipEventDestroy(ev_main);
      if (gexec) hipGraphExecDestroy(gexec);
      if (d_ctr) hipFree(d_ctr);
      if (s_wall) hipStreamDestroy(s_wall);
      if (s_wall2) hipStreamDestroy(s_wall2);
      if (ev_wall0) hipEventDestroy(ev_wall0);
      if (ev_wall) hipEventDestroy(ev_wall);
      if (ev_wall2) hipEventDestroy(ev_wall2);
      if (s_main) hipStreamDestroy(s_main);
      if (s_edge) hipStreamDestroy(s_edge);
      u0 = u1 = nullptr; s_main = s_edge = s_wall = s_wall2 = nullptr; ev_wall0 = ev_wall = ev_wall2 = nullptr;
   }

   // file-layout linear index -> padded index
   // file-layout linear index -> storage coordinates
   inline void decode(int64_t ii, int64_t &ix, int64_t &iy, int64_t &iz) const {
      const int64_t fz = ii % fNz, fy = (ii / fNz) % fNy, fx = ii / (fNz * fNy);
      ix = swz ? fz : fx; iy = fy; iz = swz ? fx : fz;
   }
   inline int64_t pad_idx(int64_t ii) const {
      int64_t ix, iy, iz;
      decode(ii, ix, iy, iz);
      return (ix * Ny + iy) * P + iz;
   }

   template <typename T> int upload(T **dst, const T *src, int64_t n) {
      *dst = nullptr;
      HIPCHK(hipMalloc((void **)dst, std::max<int64_t>(n, 1) * sizeof(T)));
      if (n > 0) HIPCHK(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
      return PF_OK;
   }
   template <typename T> int dzalloc(T **dst, int64_t n) {
      *dst = nullptr;
      size_t bytes = std::max<int64_t>(n, 1) * sizeof(T);
      { // (a full device is the one failure a caller can act on: say how much was asked for and how much there was)
         const hipError_t e = hipMalloc((void **)dst, bytes);
         if (e != hipSuccess) {
            size_t fr = 0, tot = 0;
            (void)hipGetLastError();
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) { fr = tot = 0; (void)hipGetLastError(); }
            return set_err(PF_ERR_HIP, "HIP error %s allocating %zu bytes of engine state (%zu of %zu bytes free on device %d): %s", hipGetErrorName(e), bytes, fr, tot, op.device, hipGetErrorString(e));
         }
      }
      HIPCHK(hipMemset(*dst, 0, bytes));
      return PF_OK;
   }

   // optional memory (temporal-blocking spares, autotune scratch): null instead of an error when the device is full
   template <typename T> T *try_dzalloc(int64_t n) {
      T *p = nullptr;
      const size_t bytes = std::max<int64_t>(n, 1) * sizeof(T);
      if (hipMalloc((void **)&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
      if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipGetLastError(); hipFree(p); return nullptr; }
      return p;
   }
   // split a sorted padded-index list into the ranges of plane 1 / planes 2..Nx-3 / plane Nx-2
   void plane_ranges(const std::vector<int64_t> &idx, Range &lo, Range &mid, Range &hi, int w = 1) const {
      const int64_t n = (int64_t)idx.size();
      auto first_ge = [&](int64_t px) { return (int64_t)(std::lower_bound(idx.begin(), idx.end(), px * plane) - idx.begin()); };
      if (w >= 2 && Nx < 4 * w) { lo = mid = hi = {0, 0}; return; } // (pairs / triples need far thicker slabs anyway)
      const int64_t b1 = first_ge(1), b2 = first_ge(1 + w), b3 = first_ge(Nx - 1 - w), b4 = first_ge(Nx - 1);
      lo = {b1, std::min(b2, b4)};
      if (Nx - 2 > 1) { mid = {b2, std::max(b2, b3)}; hi = {std::max(b2, b3), b4}; }
      else { mid = {b2, b2}; hi = {b2, b2}; }
      (void)n;
   }

   // Preconditions of the fused interior kernel (pf_air_fused.h).  They hold for every scene the reference's own
   // voxelizer produces (CartGrid offset 3.5 keeps walls >= 3 cells inside, sim_setup.py:91) but not for arbitrary
   // hand-made inputs, which then take the unfused kernel sequence.
   bool fused_ok() const {
      if (!use_dpp) return false;
      if (Nx < 5 || Ny < 5 || Nz < 5) return false;
      if (plane >= ((int64_t)1 << 31)) return false;   // 32-bit in-plane offsets
      // boundary nodes must not sit in the ABC shell (the reference applies ABC before the rigid update there)
      for (int64_t i = 0; i < Nb; i++) {
         int64_t ix, iy, iz;
         decode(sd.bn_ixyz[i], ix, iy, iz);
         if ((op.slab_first && ix == 1) || (op.slab_last && ix == Nx - 2) || iy == 1 || iz == 1 || iz == Nz - 2) return false;
         if (!fold && iy == Ny - 2) return false;
      }
      // receivers must not read ghost cells (their memory copy is not maintained)
      for (int64_t i = 0; i < Nr; i++) {
         int64_t ix, iy, iz;
         decode(sd.out_ixyz[i], ix, iy, iz);
         if (ix < 1 || iy < 1 || iz < 1 || ix > Nx - 2 || iy > Ny - 2 || iz > Nz - 2) return false;
      }
      // the ABC list must be the canonical shell (it is generated by the loader; a caller could pass anything)
      int64_t Nyf = fold ? 2 * (Ny - 1) : Ny;
      int64_t expect = 2 * (Nx * Nyf + Nx * Nz + Nyf * Nz) - 12 * (Nx + Nyf + Nz) + 56;
      if (fcc) expect /= 2;
      if (!(op.slab_first && op.slab_last)) return Nba <= expect; // slabs carry their share of it
      return Nba == expect;
   }
   // with a separate rigid kernel the boundary nodes read ghost MEMORY: only the folded ghost row can be adjacent
   bool rigid_separable() const {
      if (!fold) return true;
      for (int64_t i = 0; i < Nb; i++)
         if ((sd.bn_ixyz[i] / fNz) % fNy == fNy - 2) return false;
      return true;
   }

   // Store the grid with the file's x and z axes exchanged?  debug 0x1000 forces it, 0x2000 forbids it; otherwise single-domain
   // engines that own their grids decide per scene (pf__axis_exchange_pays).  Forced on a slab engine (pf_multi.hip cuts such a
   // chain along FILE Z, so that the slab axis is the storage's plane axis and ghost planes stay contiguous) its caller-owned
   // grids must hold pf_grid_bytes(Nz, Ny, Nx): planes of Ny rows of pitch(Nx).
   int decide_swap() {
      swz = false;
      const bool single = op.slab_first && op.slab_last, ext = op.ext_u0 && op.ext_u1;
      const int vb = op.air_variant & 255;
      if (op.debug & 0x1000) {
         if (op.energy) return set_err(PF_ERR_ARG, "debug 0x1000 (axes exchanged in storage): no energy diagnostic");
         swz = true;
         return PF_OK;
      }
      if ((op.debug & 0x2000) || !single || ext || op.energy || vb == 41) return PF_OK;
      int64_t counts[2];
      swz = pf__axis_exchange_pays(&sd, counts) != 0;
      if (counts[0] + counts[1] > 0 && getenv("PFFDTD_VERBOSE") && atoi(getenv("PFFDTD_VERBOSE")) > 0)
         fprintf(stderr, "pffdtd_hip: %ld of %ld boundary nodes have their successor along file x, %ld along file z: storage %s\n", (long)counts[0],
                 (long)sd.Nb, (long)counts[1], swz ? "with the x and z axes exchanged (unit stride along file x)" : "in file order");
      return PF_OK;
   }

   int init(const pf_simdata *s, const pf_opts *o) {
      sd = *s;
      op = *o;
      fNx = sd.Nx; fNy = sd.Ny; fNz = sd.Nz;
      Nx = sd.Nx; Ny = sd.Ny; Nz = sd.Nz;
      Nb = sd.Nb; Nbl = sd.Nbl; Nba = sd.Nba; Ns = sd.Ns; Nr = sd.Nr; Nt = sd.Nt;
      if (Nx < 3 || Ny < 3 || Nz < 3) return set_err(PF_ERR_ARG, "grid must be at least 3x3x3 (got %ld %ld %ld)", (long)Nx, (long)Ny, (long)Nz);
      if (sd.Npts != Nx * Ny * Nz) return set_err(PF_ERR_ARG, "Npts != Nx*Ny*Nz");
      if (sd.fcc_flag < 0 || sd.fcc_flag > 2) return set_err(PF_ERR_ARG, "fcc_flag must be 0, 1 or 2");
      if (sd.NN != (sd.fcc_flag ? 12 : 6)) return set_err(PF_ERR_ARG, "NN does not match fcc_flag");
      if (sd.Nm > PF_MNM) return set_err(PF_ERR_ARG, "too many materials (MNm=%d)", PF_MNM);
      if (Nt < 0 || Ns < 0 || Nr < 0 || Nb < 0 || Nbl < 0 || Nba < 0) return set_err(PF_ERR_ARG, "negative count");
      for (int k = 0; k < sd.Nm; k++)
         if (sd.Mb[k] < 0 || sd.Mb[k] > PF_MMB) return set_err(PF_ERR_ARG, "Mb[%d] out of range (MMb=%d)", k, PF_MMB);
      for (int k = 0; k < sd.Nm; k++) mb_max = std::max(mb_max, (int)sd.Mb[k]);
      fcc = sd.fcc_flag > 0;
      fold = sd.fcc_flag == 2;
      a1 = (Real)sd.a1; a2 = (Real)sd.a2; sl2 = (Real)sd.sl2; lo2 = (Real)sd.lo2; l = (Real)sd.l;
      { int rc = decide_swap(); if (rc) return rc; }
      if (swz) std::swap(Nx, Nz);
      P = grid_pitch(Nz, sizeof(Real));
      plane = Ny * P;
      npad = Nx * plane;

      int ndev = 0;
      if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_err(PF_ERR_NODEV, "no HIP device visible");
      if (op.device < 0 || op.device >= ndev) return set_err(PF_ERR_ARG, "device %d out of range (%d visible)", op.device, ndev);
      HIPCHK(hipSetDevice(op.device));
      int lo_prio = 0, hi_prio = 0;
      hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
      HIPCHK(hipStreamCreateWithPriority(&s_main, hipStreamNonBlocking, lo_prio));
      HIPCHK(hipStreamCreateWithPriority(&s_edge, hipStreamNonBlocking, hi_prio));
      HIPCHK(hipEventCreateWithFlags(&ev_pre, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&ev_edge, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&ev_main, hipEventDisableTiming));
      use_dpp = check_dpp(s_main) == 1;
      if (!use_dpp) return set_err(PF_ERR_HIP, "DPP wave-shift self-test failed on device %d: this library is built for gfx950 (wave64, row_shr / row_shl with bank masks)", op.device);

      // ---- state grids ----
      if (op.ext_u0 && op.ext_u1) {
         u0 = (Real *)op.ext_u0; u1 = (Real *)op.ext_u1; own_grids = false;
      } else {
         int rc;
         if ((rc = dzalloc(&u0, npad))) return rc;
         if ((rc = dzalloc(&u1, npad))) return rc;
         own_list.push_back(u0); own_list.push_back(u1);
      }

      // ---- sorted, re-based node lists ----
      auto sorted_perm = [&](const int64_t *src, int64_t n, std::vector<int64_t> &idx) { // by STORAGE index (= file order unless swz)
         std::vector<int64_t> perm(n), key(n);
         std::iota(perm.begin(), perm.end(), 0);
         for (int64_t i = 0; i < n; i++) key[i] = pad_idx(src[i]);
         bool is_sorted = true;
         for (int64_t i = 1; i < n && is_sorted; i++) is_sorted = key[i - 1] <= key[i];
         if (!is_sorted) std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return key[a] < key[b]; });
         idx.resize(n);
         for (int64_t i = 0; i < n; i++) idx[i] = key[perm[i]];
         return perm;
      };
      auto in_interior = [&](const int64_t *src, int64_t n, const char *what) -> int {
         for (int64_t i = 0; i < n; i++) {
            const int64_t ii = src[i];
            if (ii < 0 || ii >= sd.Npts) return set_err(PF_ERR_ARG, "%s[%ld]=%ld outside the grid", what, (long)i, (long)ii);
            int64_t ix, iy, iz;
            decode(ii, ix, iy, iz);
            if (ix < 1 || iy < 1 || iz < 1 || ix > Nx - 2 || iy > Ny - 2 || iz > Nz - 2)
               return set_err(PF_ERR_ARG, "%s[%ld]=%ld is not an interior node", what, (long)i, (long)ii); // fdtd_common.h:83-101
         }
         return PF_OK;
      };
      int rc;
      if ((rc = in_interior(sd.bn_ixyz, Nb, "bn_ixyz"))) return rc;
      if ((rc = in_interior(sd.bnl_ixyz, Nbl, "bnl_ixyz"))) return rc;
      if ((rc = in_interior(sd.bna_ixyz, Nba, "bna_ixyz"))) return rc;
      if ((rc = in_interior(sd.in_ixyz, Ns, "in_ixyz"))) return rc;
      for (int64_t i = 0; i < Nr; i++)
         if (sd.out_ixyz[i] < 0 || sd.out_ixyz[i] >= sd.Npts) return set_err(PF_ERR_ARG, "out_ixyz[%ld] outside the grid", (long)i);

      std::vector<int64_t> idx;
      { // boundary nodes + adjacency
         auto perm = sorted_perm(sd.bn_ixyz, Nb, idx);
         std::vector<uint16_t> adj(Nb);
         for (int64_t i = 0; i < Nb; i++) adj[i] = sd.adj_bn[perm[i]];
         if ((rc = upload(&d_bn, idx.data(), Nb))) return rc;
         if ((rc = upload(&d_adj, adj.data(), Nb))) return rc;
         plane_ranges(idx, bn_lo, bn_mid, bn_hi);
         plane_ranges(idx, bn_lo2, bn_mid2, bn_hi2, 2);
         plane_ranges(idx, bn_lo3, bn_mid3, bn_hi3, 3);
         // which interior path?  0 = automatic; 3 = the reference's kernel sequence (memory flips, marching kernel, ABC list
         // kernels); 4 = barrier-free marching kernel with virtual ghost shell + in-kernel ABC; 7 = the same with the flips in
         // memory (the 13-point default); 25 = lean fused kernel (7-point); 40 / 41 = temporally blocked pairs forced / driver only
         vbase = op.air_variant & 255;
         if (op.air_variant & ~(255 | 256)) return set_err(PF_ERR_ARG, "air_variant %d: unknown flag bits", op.air_variant);
         if (vbase != 0 && vbase != 3 && vbase != 4 && vbase != 7 && vbase != 25 && vbase != 40 && vbase != 41)
            return set_err(PF_ERR_ARG, "air_variant %d: choose 0 (auto), 3 (unfused reference sequence), 4 / 7 (barrier-free kernel: virtual ghosts / "
                                       "in-kernel ABC), 25 (lean fused kernel, 7-point), 40 / 41 (blocked pairs); the other variants were retired", op.air_variant);
         if (op.numerics != PF_NUM_CPU_EXACT && op.numerics != PF_NUM_GPU_SAFEGUARDED)
            return set_err(PF_ERR_ARG, "numerics must be PF_NUM_CPU_EXACT (0) or PF_NUM_GPU_SAFEGUARDED (2)");
         sg = op.numerics == PF_NUM_GPU_SAFEGUARDED;
         if (sg && !use_dpp) return set_err(PF_ERR_ARG, "the safeguarded numerics run on the DPP builds of the kernels only");
         const bool ok = fused_ok();
         // narrow rows (most of the last 256-column segment idle): the barrier-free kernel loses less to the idle lanes
         const int64_t Wseg = 64 * pf::VecOf<Real>::V;
         const bool wide = (double)P / (double)(cdiv(P, Wseg) * Wseg) >= 0.8;
         if (op.energy) { if (vbase != 0 && vbase != 3) return set_err(PF_ERR_ARG, "the energy diagnostic runs the unfused kernel sequence (air_variant 0 or 3)"); }
         else if (vbase == 0 || vbase == 40 || vbase == 41) {
            if ((vbase == 40 || vbase == 41) && !ok) return set_err(PF_ERR_ARG, "air_variant %d (blocked pairs) requested but the fused-path preconditions do not hold", op.air_variant);
            lean = ok && !fcc && (wide || vbase != 0);
            abck = ok && fcc;          // 13-point: flips stay in memory, the ABC loss moves into the interior kernel
            vg = ok && !lean && !fcc; // (13-point: the ghost patches on 3x(R+2) rows cost more than the flip kernels they replace)
         }
         else if (vbase == 7) { abck = true; if (!ok) return set_err(PF_ERR_ARG, "air_variant 7 (in-kernel ABC) requested but its preconditions do not hold"); }
         else if (vbase == 4) { vg = true; if (!ok) return set_err(PF_ERR_ARG, "air_variant 4 (virtual ghost shell) requested but its preconditions do not hold"); }
         else if (vbase == 25) {
            lean = true;
            if (fcc) return set_err(PF_ERR_ARG, "air_variant 25 (lean fused kernel) is 7-point Cartesian only");
            if (!ok) return set_err(PF_ERR_ARG, "air_variant 25 (lean fused kernel) requested but its preconditions do not hold");
         }
         need_fold_row = fold && !rigid_separable();
         HIPCHK(hipDeviceSynchronize()); // memsets above ran on the null stream; our streams are non-blocking
         lean_nzt = (int)cdiv(P, 64 * pf::VecOf<Real>::V);
         // skip-mask: ghost z / pad / parity, then the boundary nodes
         if ((rc = dzalloc(&mask, npad / 8))) return rc;
         HIPCHK(hipDeviceSynchronize());
         hipLaunchKernelGGL(pf::k_mask_init, dim3((unsigned)cdiv(Nx * Ny * (P / 16), 256)), dim3(256), 0, s_main, mask, Nx, Ny, P, Nz,
                            sd.fcc_flag == 1 ? 1 + (op.x_global0 & 1) : 0);
         if (Nb) hipLaunchKernelGGL(pf::k_mask_set, dim3((unsigned)cdiv(Nb, 256)), dim3(256), 0, s_main, mask, d_bn, Nb);
         HIPCHK(hipGetLastError());
      }
      { // lossy nodes
         auto perm = sorted_perm(sd.bnl_ixyz, Nbl, idx);
         std::vector<Real> ssaf(Nbl);
         std::vector<int8_t> mat(Nbl);
         for (int64_t i = 0; i < Nbl; i++) {
            ssaf[i] = ((const Real *)sd.ssaf_bnl)[perm[i]];
            mat[i] = sd.mat_bnl[perm[i]];
            if (mat[i] < 0 || mat[i] >= sd.Nm) return set_err(PF_ERR_ARG, "mat_bnl[%ld]=%d out of range", (long)perm[i], mat[i]);
         }
         if ((rc = upload(&d_bnl, idx.data(), Nbl))) return rc;
         if ((rc = upload(&d_ssaf, ssaf.data(), Nbl))) return rc;
         if ((rc = upload(&d_mat, mat.data(), Nbl))) return rc;
         plane_ranges(idx, bnl_lo, bnl_mid, bnl_hi);
         plane_ranges(idx, bnl_lo2, bnl_mid2, bnl_hi2, 2);
         plane_ranges(idx, bnl_lo3, bnl_mid3, bnl_hi3, 3);
         for (int i = 0; i < 3; i++) if ((rc = dzalloc(&ub[i], Nbl))) return rc;
         if ((rc = dzalloc(&vh1, round_up(Nbl, 64) * PF_MMB))) return rc; // [node / 64][branch][node % 64], pf::st_idx
         if ((rc = dzalloc(&gh1, round_up(Nbl, 64) * PF_MMB))) return rc;
         const int64_t nm = std::max<int64_t>(sd.Nm, 1);
         if ((rc = upload(&d_mq, (const pf::MatQuadT<Real> *)sd.mat_quads, sd.Nm ? nm * PF_MMB : 0))) return rc;
         if ((rc = upload(&d_beta, (const Real *)sd.mat_beta, sd.Nm))) return rc;
         if ((rc = upload(&d_Mb, sd.Mb, sd.Nm))) return rc;
      }
      { // fused boundary pass: map every boundary node to its lossy slot (both lists are sorted by padded index)
         std::vector<int64_t> hb(Nb), hl(Nbl);
         if (Nb) HIPCHK(hipMemcpy(hb.data(), d_bn, Nb * sizeof(int64_t), hipMemcpyDeviceToHost));
         if (Nbl) HIPCHK(hipMemcpy(hl.data(), d_bnl, Nbl * sizeof(int64_t), hipMemcpyDeviceToHost));
         std::vector<int32_t> lz(Nb, -1);
         int64_t j = 0;
         bool subset = Nbl < ((int64_t)1 << 31);
         for (int64_t i = 0; i < Nb && j < Nbl; i++) {
            if (hb[i] == hl[j]) {
               if (j + 1 < Nbl && hl[j + 1] == hl[j]) { subset = false; break; } // duplicate lossy entries: keep the separate kernels
               lz[i] = (int32_t)j++;
            } else if (hb[i] > hl[j]) { subset = false; break; }
         }
         if (j != Nbl) subset = false; // a lossy node that is not a boundary node: cannot fuse
         for (int64_t i = 1; i < Nb && subset; i++) if (hb[i] == hb[i - 1]) subset = false;
         fuse_boundary = subset && Nb > 0 && !(op.air_variant & 256) && !op.energy;
         if (fuse_boundary) { if ((rc = upload(&d_lossy, lz.data(), Nb))) return rc; }
      }
      { // ABC nodes
         auto perm = sorted_perm(sd.bna_ixyz, Nba, idx);
         std::vector<int8_t> Q(Nba);
         for (int64_t i = 0; i < Nba; i++) Q[i] = sd.Q_bna[perm[i]];
         if ((rc = upload(&d_bna, idx.data(), Nba))) return rc;
         if ((rc = upload(&d_Q, Q.data(), Nba))) return rc;
         if ((rc = dzalloc(&u2ba, Nba))) return rc;
         plane_ranges(idx, bna_lo, bna_mid, bna_hi);
      }
      { // sources: rows permuted with the nodes, samples cast to Real once (cpu_engine.h:312 casts per step)
         auto perm = sorted_perm(sd.in_ixyz, Ns, idx);
         std::vector<Real> sig((size_t)std::max<int64_t>(Ns * Nt, 1));
         for (int64_t i = 0; i < Ns; i++)
            for (int64_t n = 0; n < Nt; n++) sig[i * Nt + n] = (Real)sd.in_sigs[perm[i] * Nt + n];
         if ((rc = upload(&d_in, idx.data(), Ns))) return rc;
         if ((rc = upload(&d_insig, sig.data(), Ns * Nt))) return rc;
         plane_ranges(idx, in_lo, in_mid, in_hi);
         plane_ranges(idx, in_lo2, in_mid2, in_hi2, 2);
         plane_ranges(idx, in_lo3, in_mid3, in_hi3, 3);
      }
      { // receivers
         auto perm = sorted_perm(sd.out_ixyz, Nr, idx);
         out_row = perm;
         if ((rc = upload(&d_out, idx.data(), Nr))) return rc;
         ring_depth = op.readout_chunk > 0 ? op.readout_chunk : 1024;
         if (Nt > 0) ring_depth = std::min<int64_t>(ring_depth, Nt);
         ring_depth = std::max<int64_t>(ring_depth, 1);
         if ((rc = dzalloc(&ring, Nr * ring_depth))) return rc;
         HIPCHK(hipHostMalloc((void **)&h_ring, std::max<int64_t>(Nr * ring_depth, 1) * sizeof(Real), hipHostMallocDefault));
      }
      std::lock_guard<std::mutex> tune_lock(g_tune_mu[op.device & 63]);
      { int rc = init_tb2(); if (rc) return rc; }
      tb2_probe = true;
      // pairs or single steps?  A first measurement on the grids as allocated drops pairs that are hopeless (rooms whose clean
      // tiles are few: CTK, Musikverein) before any placement search is spent on them -- placement is worth up to ~10 %, so a
      // pair path more than 12 % behind the single steps cannot win; the survivors get their grids placed and are measured again
      pair_margin = 1.12f;
      { int rc = autotune(); if (rc) { tb2_probe = false; return rc; } }
      { int rc = sample_placement(); if (rc) { tb2_probe = false; return rc; } }
      pair_margin = 0.99f;
      if (tb2) { int rc = autotune(); if (rc) { tb2_probe = false; return rc; } }
      else if (fcc) { int rc = autotune_fcc_lw(); if (rc) { tb2_probe = false; return rc; } } // (pairs dropped or never offered)
      tb2_probe = false;
      if (tb3) tb3_remember_home();
      if (!tb2 && op.slab_first && op.slab_last) { int rc = sample_placement_single(); if (rc) return rc; }
      // hipGraph replay of the step loop (six steps per graph): measured on MI355X / ROCm 7.2 it does not beat plain
      // launches even on launch-bound grids (234x154x85: 0.0503 vs 0.0473 ms/step, 256^3: 0.0951 vs 0.0921) -- the gaps
      // between dependent kernels are the same inside a graph, and the counter-tick node adds one -- so it is opt-in
      // (debug 0x800000), kept bit-identical by the tests.
      graph_ok = (op.debug & 0x800000) && op.slab_first && op.slab_last && !tb2 && !op.timing && !op.energy;
      if (getenv("PFFDTD_VERBOSE") && atoi(getenv("PFFDTD_VERBOSE")) > 0)
         fprintf(stderr, "pffdtd_hip: engine on device %d, %ldx%ldx%ld %s %s, interior path: %s%s, numerics: %s, %d-lane row segments\n", op.device, (long)Nx, (long)Ny, (long)Nz,
                 fcc ? "13-point" : "7-point", sizeof(Real) == 4 ? "fp32" : "fp64",
                 tb3 ? ((wl_ns3 && wl_ns3z) ? "three steps per pass (k_tb3), the shell too: three-step wall regions + bricks" :
                        wl_ns3 ? "three steps per pass (k_tb3), shell: three-step x / y regions + bricks, column strips two steps + one" :
                        wl_nbrk ? "three steps per pass (k_tb3), shell: wall regions two steps + one, frame as bricks" :
                                  "three steps per pass (k_tb3), shell as wall regions + one single step") : tb2 ? "temporally blocked pairs" : (lean ? "lean fused kernel" : (vg ? "barrier-free kernel, virtual ghosts" : (abck ? "barrier-free kernel, in-kernel ABC" : "unfused reference sequence"))),
                 tb2_geom && !tb2 ? " (pairs when the caller hands over four grids)" : (swz ? " (stored with the file's x and z axes exchanged)" : ""), sg ? "GPU-safeguarded" : "CPU-exact", tb2 ? tb_lw : (lean ? 64 : pick_lw()));
      HIPCHK(hipDeviceSynchronize());
      return PF_OK;
   }

#include "pf_engine_blocking.inc"
#include "pf_engine_walls.inc"
#include "pf_engine_blocked_steps.inc"
#include "pf_engine_tune.inc"
#include "pf_engine_launch.inc"
   // ---- graph replay of the single-stream step loop ----
   int build_graph() {
      if (!d_ctr) HIPCHK(hipMalloc((void **)&d_ctr, 2 * sizeof(int64_t)));
      hipGraph_t g = nullptr;
      HIPCHK(hipStreamBeginCapture(s_main, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < 6; k++) { // the launches of step_single, with the device counters instead of n / ring_fill
         fold_x0 = 0; fold_x1 = (int)Nx;
         launch_pre(s_main);
         launch_air(s_main, 1, (int)Nx - 1);
         launch_abc(s_main, {0, Nba});
         launch_rigid(s_main, {0, Nb});
         launch_fd(s_main, {0, Nbl});
         launch_io(s_main, 0, true, {0, Ns}, d_ctr);
         hipLaunchKernelGGL(pf::k_ctr_tick, dim3(1), dim3(1), 0, s_main, d_ctr);
         rotate();
      }
      rot_count -= 6; // nothing ran: the six rotations above only walked the pointers through one period
      const hipError_t e = hipStreamEndCapture(s_main, &g);
      if (e != hipSuccess || !g) { graph_ok = false; (void)hipGetLastError(); return PF_OK; } // capture unsupported: plain launches
      if (hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0) != hipSuccess) { gexec = nullptr; graph_ok = false; (void)hipGetLastError(); }
      hipGraphDestroy(g);
      g_rot0 = ((rot_count % 6) + 6) % 6;
      return PF_OK;
   }
   int step_six(int64_t n) {
      hipLaunchKernelGGL(pf::k_ctr_set, dim3(1), dim3(1), 0, s_main, d_ctr, n, ring_fill);
      HIPCHK(hipGraphLaunch(gexec, s_main));
      for (int k = 0; k < 6; k++) rotate();
      if (ring_fill == 0) ring_n0 = n;
      ring_fill += 6;
      steps_done += 6;
      if (ring_fill == ring_depth) return flush();
      return PF_OK;
   }
   void rotate() {
      std::swap(u0, u1);
      Real *t = ub[2]; ub[2] = ub[1]; ub[1] = ub[0]; ub[0] = t;
      rot_count++;
   }
   int after_step(int64_t n) {
      if (ring_fill == 0) ring_n0 = n;
      ring_fill++;
      steps_done++;
      if (ring_fill == ring_depth) return flush();
      return PF_OK;
   }

   // one whole step on the main stream, in the reference CPU engine's order (cpu_engine.h:127-326)
   int step_single(int64_t n) {
      if (n < 0 || n >= Nt) return set_err(PF_ERR_ARG, "step %ld outside [0,Nt=%ld)", (long)n, (long)Nt);
      std::pair<hipEvent_t, hipEvent_t> ev{};
      if (op.timing) {
         if (!ev_pool.empty()) { ev = ev_pool.back(); ev_pool.pop_back(); }
         else { hipEventCreate(&ev.first); hipEventCreate(&ev.second); }
         hipEventRecord(ev.first, s_main);
      }
      fold_x0 = 0; fold_x1 = (int)Nx;
      launch_pre(s_main);
      // (The fused interior kernels skip the boundary nodes' cells and the boundary pass reads u^n only, so the two commute --
      // but running the pass beside the interior kernel on the second stream gains nothing on the rooms: Musikverein 3.87 vs
      // 3.82 ms per step one after the other, CTK 0.547-0.563 vs 0.556-0.558, round 4.)
      launch_air(s_main, 1, (int)Nx - 1);
      launch_abc(s_main, {0, Nba});
      launch_rigid(s_main, {0, Nb});
      launch_fd(s_main, {0, Nbl});
      launch_io(s_main, n, true, {0, Ns});
      if (op.timing) { hipEventRecord(ev.second, s_main); step_ev.push_back(ev); }
      HIPCHK(hipGetLastError());
      rotate();
      return after_step(n);
   }

   // ---------------- energy diagnostic (python/fdtd/sim_fdtd.py:587-620) ----------------
   int energy_cfg(double h, double c, double Ts, const double *DEF) override {
      if (!op.energy) return set_err(PF_ERR_STATE, "engine was not created with pf_opts.energy=1");
      if (sd.fcc_flag == 2) return set_err(PF_ERR_ARG, "the energy diagnostic is defined for fcc_flag 0 and 1 (as in the reference)");
      HIPCHK(hipSetDevice(op.device));
      en_h = h; en_c = c; en_Ts = Ts;
      int rc;
      if (!Lu) {
         if ((rc = dzalloc(&Lu, npad))) return rc;
         if ((rc = dzalloc(&vh_old, round_up(Nbl, 64) * PF_MMB))) return rc;
         if ((rc = dzalloc(&u2in, Ns))) return rc;
         if ((rc = dzalloc(&d_acc, (int64_t)pf::EN_NACC))) return rc;
         if ((rc = upload(&d_DEF, DEF, (int64_t)std::max<int>(sd.Nm, 1) * PF_MMB * 3))) return rc;
      }
      HIPCHK(hipDeviceSynchronize());
      en_ready = true;
      return PF_OK;
   }
   int run_energy(int64_t n0, int64_t nsteps, double *H, double *El, double *Ei) override {
      if (!en_ready) return set_err(PF_ERR_STATE, "call pf_engine_energy_cfg first");
      if (in_step) return set_err(PF_ERR_STATE, "pf_engine_run_energy inside a split-phase step");
      HIPCHK(hipSetDevice(op.device));
      hipStream_t s = s_main;
      const double V = fcc ? 2.0 : 1.0, l2d = sd.l2, ld = sd.l;
      const dim3 g3((unsigned)cdiv(Nz, 256), (unsigned)(Ny - 2), (unsigned)(Nx - 2));
      auto g1 = [](int64_t n, int b) { return dim3((unsigned)std::max<int64_t>(cdiv(n, b), 1)); };
      for (int64_t n = n0; n < n0 + nsteps; n++) {
         if (n < 0 || n >= Nt) return set_err(PF_ERR_ARG, "step %ld outside [0,Nt=%ld)", (long)n, (long)Nt);
         double acc[pf::EN_NACC];
         HIPCHK(hipMemsetAsync(d_acc, 0, sizeof(double) * pf::EN_NACC, s));
         // state before the step: u0 = u^{n-1} (u2), u1 = u^n, Lu = L(u^{n-1})
         hipLaunchKernelGGL(pf::k_energy_int<Real>, g3, dim3(256), 0, s, u1, u0, Lu, Nx, Ny, Nz, P, plane, l2d, d_acc);
         if (Nba) hipLaunchKernelGGL(pf::k_energy_abc<Real>, g1(Nba, 256), dim3(256), 0, s, u1, u0, Lu, d_bna, d_Q, Nba, l2d, d_acc);
         if (Nbl) hipLaunchKernelGGL(pf::k_energy_stored<Real>, g1(Nbl, 256), dim3(256), 0, s, vh1, gh1, d_ssaf, d_mat, d_Mb, d_DEF, Nbl, en_Ts, d_acc);
         if (Ns) hipLaunchKernelGGL(pf::k_energy_in<Real>, g1(Ns, 64), dim3(64), 0, s, u0, u2in, d_in, d_insig, Ns, Nt, n, 0, d_acc);
         if (Nbl) HIPCHK(hipMemcpyAsync(vh_old, vh1, sizeof(Real) * round_up(Nbl, 64) * PF_MMB, hipMemcpyDeviceToDevice, s));
         // the step itself (unfused sequence), with Lu = L(u1) taken after the ghost flips
         fold_x0 = 0; fold_x1 = (int)Nx;
         launch_pre(s);
         if (fcc) {
            hipLaunchKernelGGL((pf::k_lap_air<Real, true>), g3, dim3(256), 0, s, u1, Lu, mask, Nx, Ny, Nz, P, plane);
            if (Nb) hipLaunchKernelGGL((pf::k_lap_bn<Real, true>), g1(Nb, 256), dim3(256), 0, s, u1, Lu, d_bn, d_adj, P, plane, Nb);
         } else {
            hipLaunchKernelGGL((pf::k_lap_air<Real, false>), g3, dim3(256), 0, s, u1, Lu, mask, Nx, Ny, Nz, P, plane);
            if (Nb) hipLaunchKernelGGL((pf::k_lap_bn<Real, false>), g1(Nb, 256), dim3(256), 0, s, u1, Lu, d_bn, d_adj, P, plane, Nb);
         }
         launch_air(s, 1, (int)Nx - 1);
         launch_abc(s, {0, Nba});
         launch_rigid(s, {0, Nb});
         launch_fd(s, {0, Nbl});
         launch_io(s, n, true, {0, Ns});
         // after the step (u0 = u^{n+1} until the rotation)
         if (Nbl) hipLaunchKernelGGL(pf::k_energy_loss<Real>, g1(Nbl, 256), dim3(256), 0, s, vh_old, vh1, d_ssaf, d_mat, d_Mb, d_DEF, Nbl, d_acc);
         if (Nba) hipLaunchKernelGGL(pf::k_energy_abcloss<Real>, g1(Nba, 256), dim3(256), 0, s, u0, u2ba, d_bna, d_Q, Nba, d_acc);
         if (Ns) hipLaunchKernelGGL(pf::k_energy_in<Real>, g1(Ns, 64), dim3(64), 0, s, u0, u2in, d_in, d_insig, Ns, Nt, n, 1, d_acc);
         HIPCHK(hipGetLastError());
         HIPCHK(hipMemcpyAsync(acc, d_acc, sizeof(acc), hipMemcpyDeviceToHost, s));
         HIPCHK(hipStreamSynchronize(s));
         H[n] = V * 0.5 * en_h * acc[pf::EN_INT] - V * 0.5 * en_h * acc[pf::EN_ABC] + V * 0.5 * en_c / l2d * acc[pf::EN_STORED];
         El[n + 1] = El[n] + V * 0.25 * en_h / ld * acc[pf::EN_LOSS] + 0.5 * V * en_h / ld * acc[pf::EN_ABCLOSS];
         Ei[n + 1] = Ei[n] + (V * en_h / l2d) * 0.5 * acc[pf::EN_IN];
         rotate();
         int rc = after_step(n);
         if (rc) return rc;
      }
      return flush();
   }

   int run(int64_t n0, int64_t nsteps) override {
      if (in_step || pair_phase) return set_err(PF_ERR_STATE, "pf_engine_run inside a split-phase step (pair)");
      HIPCHK(hipSetDevice(op.device));
      for (int64_t n = n0; n < n0 + nsteps;) {
         int rc;
         // temporally blocked pairs come in twos, so that the state is back in the caller's two grids afterwards
         if (tb3) tb3_pick(); // (single steps swap the state grids: the targets follow)
         if (tb3 && n + 3 <= n0 + nsteps && ring_fill + 3 <= ring_depth) {
            if ((rc = step_triple(n))) return rc;
            n += 3;
         } else if (tb3 && n + 2 <= n0 + nsteps && ring_fill + 2 <= ring_depth) {
            if ((rc = step_pair(n))) return rc; // the last two steps of a run: a pair on the triples' tiles (k_tb3's two-step form)
            n += 2;
         } else if (tb2 && !tb3 && n + 4 <= n0 + nsteps && ring_fill + 4 <= ring_depth) {
            if ((rc = step_pair(n))) return rc;
            if ((rc = step_pair(n + 2))) return rc;
            n += 4;
         } else if (graph_ok && n + 6 <= n0 + nsteps && ring_fill + 6 <= ring_depth && (gexec || g_rot0 < 0) &&
                    (g_rot0 < 0 || ((rot_count % 6) + 6) % 6 == g_rot0)) {
            if (!gexec) { if ((rc = build_graph())) return rc; if (!gexec) continue; }
            if ((rc = step_six(n))) return rc;
            n += 6;
         } else {
            if ((rc = step_single(n))) return rc;
            n++;
         }
         if (op.timing && air_ev.size() >= 512) { rc = harvest(); if (rc) return rc; }
      }
      int rc = flush();
      if (rc) return rc;
      return harvest();
   }

#include "pf_engine_slab_steps.inc"
   int sync() override {
      HIPCHK(hipSetDevice(op.device));
      HIPCHK(hipStreamSynchronize(s_edge));
      HIPCHK(hipStreamSynchronize(s_main));
      return PF_OK;
   }

   // ring -> sd.u_out[row*Nt + n]  ((double) cast as cpu_engine.h:306)
   int flush() override {
      if (ring_fill == 0) return PF_OK;
      HIPCHK(hipSetDevice(op.device));
      HIPCHK(hipStreamSynchronize(s_edge));
      if (Nr > 0) {
         HIPCHK(hipMemcpyAsync(h_ring, ring, (size_t)(Nr * ring_depth) * sizeof(Real), hipMemcpyDeviceToHost, s_main));
      }
      HIPCHK(hipStreamSynchronize(s_main));
      if (sd.u_out)
         for (int64_t t = 0; t < Nr; t++) {
            double *dst = sd.u_out + out_row[t] * Nt + ring_n0;
            const Real *src = h_ring + t * ring_depth;
            for (int64_t k = 0; k < ring_fill; k++) dst[k] = (double)src[k];
         }
      ring_fill = 0;
      return PF_OK;
   }

   int harvest() {
      if (!op.timing) return PF_OK;
      HIPCHK(hipStreamSynchronize(s_edge));
      HIPCHK(hipStreamSynchronize(s_main));
      for (auto &p : air_ev) {
         float ms = 0;
         HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
         tm.air_ms_total += ms; tm.air_launches++;
         ev_pool.push_back(p);
      }
      air_ev.clear();
      for (auto &p : tb2_ev) {
         float ms = 0;
         HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
         tm.tb2_ms_total += ms; tm.tb2_launches += 1;
         ev_pool.push_back(p);
      }
      tb2_ev.clear();
      tm.tb2_cells = tb_clean_cells;
      tm.tb_steps_per_pass = triples() ? 3 : ((tb2 || tb2_slab) ? 2 : 0);
      for (auto &p : step_ev) {
         float ms = 0;
         HIPCHK(hipEventElapsedTime(&ms, p.first, p.second));
         tm.step_ms_total += ms; tm.steps++;
         ev_pool.push_back(p);
      }
      step_ev.clear();
      return PF_OK;
   }
   int timing(pf_timing *t, int reset) override {
      int rc = harvest();
      if (rc) return rc;
      for (int i = 0; i < 3; i++) tm.tune_ms[i] = tune_ms[i];
      tm.air_path = (tb2 || tb2_slab) ? 2 : (lean ? 0 : (vg ? 1 : -1));
      tm.tb2_lw = (tb2 || tb2_slab) ? tb_lw : 0;
      tm.tb2_dirty_tiles = (tb2 || tb2_slab) ? tb_ndirty : 0;
      tm.place_candidates = (int64_t)place_ms.size();
      if (!place_ms.empty()) {
         tm.place_ms[0] = place_ms[0];
         tm.place_ms[1] = *std::min_element(place_ms.begin(), place_ms.end());
         tm.place_ms[2] = *std::max_element(place_ms.begin(), place_ms.end());
      }
      tm.wall_blocks[0] = tm.wall_blocks[1] = 0;
      if (wl_on)
         for (const WlGroup &g : wl_grp) { tm.wall_blocks[0] += g.nblk[0] + g.nblk[2]; tm.wall_blocks[1] += g.nblk[1]; }
      tm.wall_bricks = wl_on ? wl_nbrk : 0;
      tm.wall_three_steps = (wl_on && tb3) ? ((wl_ns3 ? 1 : 0) | (wl_ns3z ? 8 : 0)) : 0;
      if (t) *t = tm;
      if (reset) tm = pf_timing{};
      return PF_OK;
   }

   int set_timing(int on) override {
      if (in_step) return set_err(PF_ERR_STATE, "pf_engine_set_timing inside a step");
      int rc = harvest();
      if (rc) return rc;
      op.timing = on ? 1 : 0;
      return PF_OK;
   }
   int get_grid(int which, void *host) override {
      HIPCHK(hipSetDevice(op.device));
      int rc = sync();
      if (rc) return rc;
      const Real *src = which == 0 ? (pair_phase > 0 ? (const Real *)u0_src : (const Real *)u0) : u1; // mid-pair u0 already names the grid being written
      if ((lean || vg) && which == 1) { // write the virtual ghost shell out, exactly as the reference's flips would have
         launch_flips(s_main);
         HIPCHK(hipStreamSynchronize(s_main));
      }
      if (swz) { // storage -> file order through a device-side transposition
         Real *tmp = nullptr;
         HIPCHK(hipMalloc((void **)&tmp, (size_t)sd.Npts * sizeof(Real)));
         hipLaunchKernelGGL((pf::k_storage_to_file<Real>), dim3((unsigned)cdiv(sd.Npts, 256)), dim3(256), 0, s_main, src, tmp, fNx, fNy, fNz, Ny, P);
         const hipError_t e = hipMemcpyAsync(host, tmp, (size_t)sd.Npts * sizeof(Real), hipMemcpyDeviceToHost, s_main);
         hipStreamSynchronize(s_main);
         hipFree(tmp);
         HIPCHK(e);
         return PF_OK;
      }
      HIPCHK(hipMemcpy2D(host, Nz * sizeof(Real), src, P * sizeof(Real), Nz * sizeof(Real), Nx * Ny, hipMemcpyDeviceToHost));
      return PF_OK;
   }
   int set_grid(int which, const void *host) override {
      HIPCHK(hipSetDevice(op.device));
      int rc = sync();
      if (rc) return rc;
      Real *dst = which == 0 ? u0 : u1;
      state_touched = true;
      if (swz) {
         Real *tmp = nullptr;
         HIPCHK(hipMalloc((void **)&tmp, (size_t)sd.Npts * sizeof(Real)));
         hipError_t e = hipMemcpyAsync(tmp, host, (size_t)sd.Npts * sizeof(Real), hipMemcpyHostToDevice, s_main);
         hipLaunchKernelGGL((pf::k_file_to_storage<Real>), dim3((unsigned)cdiv(sd.Npts, 256)), dim3(256), 0, s_main, tmp, dst, fNx, fNy, fNz, Ny, P);
         hipStreamSynchronize(s_main);
         hipFree(tmp);
         HIPCHK(e);
         return PF_OK;
      }
      HIPCHK(hipMemcpy2D(dst, P * sizeof(Real), host, Nz * sizeof(Real), Nz * sizeof(Real), Nx * Ny, hipMemcpyHostToDevice));
      return PF_OK;
   }
   void *stream(int which) override { return which == 1 ? (void *)s_edge : (void *)s_main; }
};

} // namespace

struct pf_engine {
   EngineBase *impl;
};

extern "C" {

const char *pf_last_error(void) { return g_err.c_str(); }
const char *pf_version(void) { return "pffdtd_hip 0.2 (gfx950)"; }

int pf_device_count(void) {
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) return 0;
   return n;
}

int64_t pf_grid_pitch(int64_t Nz, int32_t real_bytes) {
   if (real_bytes != 4 && real_bytes != 8) return -1;
   return grid_pitch(Nz, real_bytes);
}
size_t pf_grid_bytes(int64_t Nx, int64_t Ny, int64_t Nz, int32_t real_bytes) {
   if (real_bytes != 4 && real_bytes != 8) return 0;
   return (size_t)(Nx * Ny * grid_pitch(Nz, real_bytes)) * (size_t)real_bytes;
}

void pf_opts_default(pf_opts *o) {
   if (!o) return;
   memset(o, 0, sizeof *o);
   o->slab_first = 1;
   o->slab_last = 1;
}

int pf_engine_create(const pf_simdata *sd, const pf_opts *opts, pf_engine **out) {
   if (!sd || !out) return set_err(PF_ERR_ARG, "null argument");
   *out = nullptr;
   pf_opts o;
   if (opts) o = *opts; else pf_opts_default(&o);
   EngineBase *impl = nullptr;
   int rc;
   if (sd->real_bytes == 4) {
      auto *e = new Engine<float>();
      rc = e->init(sd, &o);
      impl = e;
#ifndef PF_DEV_F32_ONLY // (development builds, PFFDTD_DEV_F32=1 in pffdtd_amd/build.py: half the compile time; never shipped)
   } else if (sd->real_bytes == 8) {
      auto *e = new Engine<double>();
      rc = e->init(sd, &o);
      impl = e;
#endif
   } else {
      return set_err(PF_ERR_ARG, "real_bytes must be 4 or 8 (got %d)", sd->real_bytes);
   }
   if (rc) { std::string keep = g_err; delete impl; g_err = keep; return rc; }
   *out = new pf_engine{impl};
   return PF_OK;
}

void pf_engine_destroy(pf_engine *e) {
   if (!e) return;
   delete e->impl;
   delete e;
}

#define PF_NEED(e) if (!(e) || !(e)->impl) return set_err(PF_ERR_ARG, "null engine")

int pf_engine_run(pf_engine *e, int64_t n0, int64_t nsteps) { PF_NEED(e); return e->impl->run(n0, nsteps); }
int pf_engine_step_begin(pf_engine *e, int64_t n) { PF_NEED(e); return e->impl->step_begin(n); }
int pf_engine_halo_ptrs(pf_engine *e, void **send_lo, void **send_hi, void **recv_lo, void **recv_hi, size_t *plane_bytes) {
   PF_NEED(e);
   return e->impl->halo_ptrs(send_lo, send_hi, recv_lo, recv_hi, plane_bytes);
}
int pf_engine_step_end(pf_engine *e, int64_t n) { PF_NEED(e); return e->impl->step_end(n); }
int pf_engine_state_grids(pf_engine *e, void **u_prev, void **u_cur) { PF_NEED(e); return e->impl->state_grids(u_prev, u_cur); }
int pf_engine_layout(pf_engine *e, int64_t *dims, int64_t *pitch, int32_t *exchanged) { PF_NEED(e); return e->impl->layout(dims, pitch, exchanged); }
int pf_engine_set_spares(pf_engine *e, void *g2, void *g3) { PF_NEED(e); return e->impl->set_spares(g2, g3); }
int pf_engine_place_grids(pf_engine *e, void *const *grids, int32_t n, int32_t *idx) { PF_NEED(e); return e->impl->place_grids(grids, n, idx); }
int pf_engine_place_grids5(pf_engine *e, void *const *grids, int32_t n, int32_t *idx) { PF_NEED(e); return e->impl->place_grids5(grids, n, idx); }
void *pf_engine_stream(pf_engine *e, int32_t which) { return (e && e->impl) ? e->impl->stream(which) : nullptr; }
int pf_engine_sync(pf_engine *e) { PF_NEED(e); return e->impl->sync(); }
int pf_engine_flush_outputs(pf_engine *e) { PF_NEED(e); return e->impl->flush(); }
int pf_engine_get_grid(pf_engine *e, int32_t which, void *host) { PF_NEED(e); return e->impl->get_grid(which, host); }
int pf_engine_set_grid(pf_engine *e, int32_t which, const void *host) { PF_NEED(e); return e->impl->set_grid(which, host); }
int pf_engine_timing(pf_engine *e, pf_timing *t, int32_t reset) { PF_NEED(e); return e->impl->timing(t, reset); }
int pf_engine_set_timing(pf_engine *e, int32_t on) { PF_NEED(e); return e->impl->set_timing(on); }
int pf_engine_energy_cfg(pf_engine *e, double h, double c, double Ts, const double *DEF) { PF_NEED(e); if (!DEF) return set_err(PF_ERR_ARG, "null DEF"); return e->impl->energy_cfg(h, c, Ts, DEF); }
int pf_engine_run_energy(pf_engine *e, int64_t n0, int64_t nsteps, double *H_tot, double *E_lost, double *E_in) {
   PF_NEED(e);
   if (!H_tot || !E_lost || !E_in) return set_err(PF_ERR_ARG, "null output array");
   return e->impl->run_energy(n0, nsteps, H_tot, E_lost, E_in);
}

} // extern "C"
