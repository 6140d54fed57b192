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
#include "pf_kernels.h"
#include "pf_air_fused.h"
#include "pf_energy.h"
#include "pf_tb2.h"
#include "pf_tb3.h"
#include "pf_wall.h"

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
      F(wl_pen); F(wl_rec); F(wl_rest); F(wl_blk); F(vh1b); F(gh1b); F(d_lossy); F(mask); F(zs_map); F(zs_adj); F(zs_li); F(zs_rest); F(zs_fd); F(tb_clean); F(tb_dirty); F(tb_sample); F(sh_tiles); F(Lu); F(vh_old); F(u2in); F(d_acc); F(d_DEF); F(d_bn); F(d_bnl); F(d_bna); F(d_in); F(d_out); F(d_adj); F(d_Q); F(d_mat); F(d_Mb); F(d_ssaf);
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
                 tb3 ? "three steps per pass (k_tb3), shell as wall regions + one single step" : tb2 ? "temporally blocked pairs" : (lean ? "lean fused kernel" : (vg ? "barrier-free kernel, virtual ghosts" : (abck ? "barrier-free kernel, in-kernel ABC" : "unfused reference sequence"))),
                 tb2_geom && !tb2 ? " (pairs when the caller hands over four grids)" : (swz ? " (stored with the file's x and z axes exchanged)" : ""), sg ? "GPU-safeguarded" : "CPU-exact", tb2 ? tb_lw : (lean ? 64 : pick_lw()));
      HIPCHK(hipDeviceSynchronize());
      return PF_OK;
   }

   // ---------------- temporal blocking: two steps per pass over the boundary-free box ----------------
   // Every boundary node of a box-shaped room sits within a few cells of a grid face; the box of cells at least two
   // cells deeper than the deepest boundary node (and off the ABC shell) sees nothing but the plain air update for two
   // consecutive steps, so k_tb2_reg may produce u^{n+1} and u^{n+2} there in one pass (16 B per cell instead of 24).
   // The shell around the box (x slabs, row strips, column strips; all boundary / ABC / source cells live there) is
   // stepped twice by the single-step kernels, out of place.  Rooms with interior geometry have no such box: tb2 stays
   // off and nothing changes.  air_variant 0 (auto) and 40 enable it, 41 = same driver with the box disabled (tests).
   // undo a temporal-blocking arrangement (its grids, tile lists, strip tables, wall regions): the engine steps singly
   void drop_blocking() {
      tb2 = tb3 = false;
      free_walls();
      for (Real **g : {&bufC, &bufD, &bufE})
         if (*g) { own_list.erase(std::remove(own_list.begin(), own_list.end(), *g), own_list.end()); hipFree(*g); *g = nullptr; }
      auto F = [](auto *&p) { if (p) hipFree((void *)p); p = nullptr; };
      F(zs_map); F(zs_adj); F(zs_li); F(zs_rest); F(zs_fd);
      zs_mode = 0;
   }
   // Three steps per pass where they can be had: single-domain 7-point engines in file-order storage, 64-lane row segments, the
   // shell as wall regions (init_walls).  The box then keeps THREE cells from anything that is not a plain air update (k_tb3
   // computes u^{n+1} two cells beyond it).  Anything else: pairs as before.  debug 0x20000: never triples.
   int init_tb2() {
      tb3 = tb3_geom = tb3_slab = false;
      const bool single = op.slab_first && op.slab_last;
      if (!fcc && !swz && !(op.debug & (0x20000 | 0x300 | 0x10000000)) && vbase != 41) {
         int rc = init_tb2_impl(true);
         if (rc) return rc;
         if (single && tb2 && wl_on) { tb3 = true; return PF_OK; }
         // slab engines: the triples' box and tiles stand; whether they are used is decided when the caller hands over its grids
         // (pf_engine_place_grids5: five grids and wall regions that fit; else the pairs' geometry is rebuilt there)
         if (!single && tb2_geom) { tb3_geom = true; return PF_OK; }
         drop_blocking();
      }
      return init_tb2_impl(false);
   }
   int init_tb2_impl(bool triple) {
      tb2 = tb2_geom = tb2_slab = false;
      const bool single = op.slab_first && op.slab_last;
      if (op.energy || (op.debug & 0x4000)) return PF_OK; // 0x4000: single steps only
      // exchanged axes (rooms): the pair kernels work in storage coordinates and take their neighbours in the FILE's order (template
      // flag SWZ: 64-lane row segments, single-domain engines); 7-point: k_tb2_reg / k_tb1_tile / k_air_zstrip<..., SWZ>
      if (swz && !single) return PF_OK;
      // 7-point: the fused single-step kernels carry the shell; 13-point: folded grids with the flips in memory and the ABC
      // loss in the interior kernel (the automatic 13-point arrangement)
      if (fcc ? !(fold && abck) : !(lean || vg)) return PF_OK;
      if (!(vbase == 0 || vbase == 40 || vbase == 41) || !use_dpp) return PF_OK;
      if (Nb > 0 && !boundary_fused()) return PF_OK;
      // Margins of the box: three cells off every grid face (the ABC cells sit at index 1 and the box must stay two cells
      // away from anything that is not a plain air update), deeper where a wall layer hugs the face -- a face whose plane at
      // depth d <= 12 is at least half boundary nodes pushes the box to depth d + 2 (shoebox rooms: walls at depth 2-3,
      // box from depth 5).  Whatever geometry remains inside the box is dealt with tile by tile below.
      const int64_t NzNy = Nz * Ny;
      int64_t hist[6][16] = {};
      for (int64_t i = 0; i < Nb; i++) {
         int64_t ix, iy, iz; // storage coordinates
         decode(sd.bn_ixyz[i], ix, iy, iz);
         const int64_t d[6] = {ix, Nx - 1 - ix, iy, Ny - 1 - iy, iz, Nz - 1 - iz};
         for (int f = 0; f < 6; f++) if (d[f] < 16) hist[f][d[f]]++;
      }
      const int reach = triple ? 3 : 2; // how far the blocked kernel's own u^{n+1} (u^{n+2}) reach beyond the box, plus one
      auto margin = [&](int f, int64_t area) {
         int m = reach + 1;
         for (int d = 0; d <= 12; d++) if (hist[f][d] * 2 >= area) m = std::max(m, d + reach);
         return m;
      };
      constexpr int V = pf::VecOf<Real>::V;
      const int hl = (triple && V < 4) ? 2 : 1; // halo lanes per side of a row segment (k_tb3 in fp64: two, pf_tb3.h)
      // towards a neighbouring slab the box stops three planes short of the ghost plane: planes 1-2 / Nx-3..Nx-2 are the
      // edge planes of a split-phase pair (plane 1 needs the neighbour's data between the two steps; with plane 2 on the
      // edge stream as well the box kernel never reads a ghost plane, so the main stream never waits for an exchange)
      // (triples: the edge stream owns three planes per side, the box stops four short of the ghost plane)
      tbx0 = op.slab_first ? margin(0, NzNy) : reach + 1; tbx1 = (int)Nx - (op.slab_last ? margin(1, NzNy) : reach + 1);
      tby0 = margin(2, Nx * Nz); tby1 = (int)Ny - margin(3, Nx * Nz);
      const int mz0 = (margin(4, Nx * Ny) + 3) / 4 * 4, mz1 = (margin(5, Nx * Ny) + 3) / 4 * 4;
      tbz0 = mz0;
      // row segments of 64 / 32 / 16 lanes (a wave stacks 1 / 2 / 4 of them in y): the width that needs the fewest lanes
      // for the box's z range (ties: the widest).  A last z tile with only a sliver of core columns costs a whole
      // workgroup per (row tile, x chunk) and re-reads lines the right column strip streams anyway: up to two 128-byte
      // lines of columns are left to the strip instead.
      {
         int64_t best = -1;
         for (int lw : {64, 32, 16}) {
            if (op.debug & 0x300) { if (lw != ((op.debug & 0x100) ? 32 : 16)) continue; } // tuning override (as pick_lw)
            else if (((op.debug & 0x400) || swz || triple) && lw != 64) continue; // (exchanged axes: the SWZ instantiations exist for 64-lane segments only; such rooms have long rows; k_tb3: 64 lanes)
            const int TC = (lw - 2 * hl) * V;
            int z1 = (int)((Nz - mz1) / 4 * 4);
            const int nz = z1 - tbz0, rem = nz % TC;
            if (nz > TC && rem > 0 && rem * (int)sizeof(Real) <= 256) z1 -= rem;
            if (z1 - tbz0 < TC / 2) continue;
            const int64_t lanes = cdiv(z1 - tbz0, TC) * lw;
            if (best < 0 || lanes < best) { best = lanes; tb_lw = lw; tbz1 = z1; }
         }
         if (best < 0) return PF_OK; // no room for a single row segment
      }
      // Wall regions (init_walls): a column strip costs one 128-byte line per row whatever its width, but its pencils live in
      // registers -- a sliver cut off the box is shared between the two strips instead of all going to the right one.
      if (!fcc && !(op.debug & 0x10000000)) { // (slab engines too: init_walls(true))
         const int z1full = (int)((Nz - mz1) / 4 * 4);
         if (tbz1 < z1full) {
            const int rem = z1full - tbz1;
            // (the shift that lets both strips become wall regions with the fewest pencils cut in two; else the most even one)
            const bool ps = (op.debug & 0x2000000) != 0 || sizeof(Real) != 4;
            int best_sh = 0, best_need = 1 << 30;
            for (int sh = 0; sh <= rem; sh += 4) {
               const int lo = wl_lo_option(tbz0 + sh, ps), hi = wl_hi_option(tbz1 + sh, ps);
               int need = std::max(tbz0 + sh + 2, (int)Nz - ((tbz1 + sh - 2) / 4 * 4));
               need += (lo && hi) ? ((lo == 2) + (hi == 2)) * 100 : 1000;
               if (need < best_need) { best_need = need; best_sh = sh; }
            }
            tbz0 += best_sh; tbz1 += best_sh;
         }
      }
      szl = tbz0; szr = tbz1; // (widened to whole 128-byte lines or 32-byte sectors, the overlap with the box computed twice: 0-2 % slower, re-measured with the grids placed)
      if (vbase == 41) tbx1 = tbx0; // driver test: everything goes through the out-of-place single-step path
      tb_xr.clear();
      // rows of a workgroup: 4 waves x R = 3 (7-point); 13-point: 6 inner waves x R = 2 with 64-lane segments (k_tb2_fcc_x), else 4 x 2
      const int TC = (tb_lw - 2 * hl) * V, TR = triple ? tb3_rows : (fcc ? ((tb_lw == 64 && fcc_wt) ? 2 * (fcc_wt - 2) : 8 * (64 / tb_lw)) : 12 * (64 / tb_lw));
      int64_t vol = 0;
      if (tbx1 - tbx0 >= 16 && tby1 - tby0 >= 24 && tbz1 - tbz0 >= TC / 2) {
         tb_xr.push_back({tbx0, tbx1});
         const int np = tbx1 - tbx0;
         // ~16-plane chunks, even split (tools/tb2_probe.py); 13-point: ~24 (3.93 vs 4.07 ms per launch at 1024^3, 32-48 the same)
         // (k_tb3: 64 -- 3.28 ms per launch at 1024^3 against 3.57 with 32 and 3.63 with 16 in the plain tile order, tools/tb3_probe.py)
         // -- but a thin slab needs enough workgroups to fill the chip: at least ~1024 tiles, chunks of 16 planes or more (a rank of 8 at
         // 1024^3: 125 box planes in 2 chunks of 63 were 408 workgroups on 256 CUs)
         int want_chunk = fcc ? 24 : tb2_chunk;
         if (triple) {
            const int64_t tiles_yz = cdiv(tby1 - tby0, TR) * cdiv(tbz1 - tbz0, TC);
            want_chunk = (int)std::min<int64_t>(64, std::max<int64_t>(16, np / std::max<int64_t>(cdiv(1024, std::max<int64_t>(tiles_yz, 1)), 1)));
         }
         tb_chunk = (int)cdiv(np, std::max<int64_t>(cdiv(np, want_chunk), 1));
         tb_nxc = (int)cdiv(np, tb_chunk); tb_nyt = (int)cdiv(tby1 - tby0, TR); tb_nzt = (int)cdiv(tbz1 - tbz0, TC);
         const int64_t ntile = (int64_t)tb_nxc * tb_nyt * tb_nzt;
         if (ntile >= ((int64_t)1 << 31)) return PF_OK;
         std::vector<uint8_t> dirty((size_t)ntile, 0);
         // every tile whose core, grown by `grow` cells, holds this cell (pairs: one cell -- the kernel's own u^{n+1} reach one cell
         // beyond the core; triples: two)
         auto mark = [&](int64_t ii, int grow) {
            int64_t ix64, iy64, iz64;
            decode(ii, ix64, iy64, iz64);
            const int ix = (int)ix64, iy = (int)iy64, iz = (int)iz64;
            auto span = [grow](int c, int org, int size, int n, int end, int &lo, int &hi) {
               if (c < org - grow || c > end - 1 + grow) return false;
               lo = (c - grow - org) >= 0 ? (c - grow - org) / size : 0;
               hi = std::min(std::max(c + grow - org, 0) / size, n - 1);
               return lo <= hi;
            };
            int x0, x1, y0, y1, z0, z1;
            if (!span(ix, tbx0, tb_chunk, tb_nxc, tbx1, x0, x1) || !span(iy, tby0, TR, tb_nyt, tby1, y0, y1) ||
                !span(iz, tbz0, TC, tb_nzt, tbz1, z0, z1)) return;
            for (int a = x0; a <= x1; a++) for (int b = y0; b <= y1; b++) for (int c = z0; c <= z1; c++)
               dirty[((size_t)a * tb_nyt + b) * tb_nzt + c] = 1;
         };
         const int grow = triple ? 2 : 1;
         for (int64_t i = 0; i < Nb; i++) mark(sd.bn_ixyz[i], grow);
         for (int64_t i = 0; i < Ns; i++) mark(sd.in_ixyz[i], grow); // (the source is added between the steps)
         if (triple) // a receiver reads u^{n+1} from memory, which a clean tile of k_tb3 never stores: its tile steps singly
            for (int64_t i = 0; i < Nr; i++) mark(sd.out_ixyz[i], 0);
         std::vector<int32_t> cl, di;
         for (int64_t t = 0; t < ntile; t++) {
            if (dirty[t]) { di.push_back((int32_t)t); continue; }
            cl.push_back((int32_t)t);
            const int zt = (int)(t % tb_nzt), yt = (int)((t / tb_nzt) % tb_nyt), xc = (int)(t / ((int64_t)tb_nzt * tb_nyt));
            vol += (int64_t)(std::min(tbx0 + (xc + 1) * tb_chunk, tbx1) - (tbx0 + xc * tb_chunk)) *
                   (std::min(tby0 + (yt + 1) * TR, tby1) - (tby0 + yt * TR)) * (std::min(tbz0 + (zt + 1) * TC, tbz1) - (tbz0 + zt * TC));
         }
         tb_order_band = true;
         if (tb_order_band && !cl.empty()) {
            // XCD-banded order: hardware places block b on XCD b % 8; each XCD gets a contiguous band of the clean tiles of
            // every x chunk (tiles that share halo rows then share an L2), blocks b .. b+7 walking the 8 bands in step.
            // Measured at 1024^3, alternating runs on one box: k_tb2_reg 3.258 vs 3.295 ms per launch, whole step 427.3 vs
            // 422.2 Gvox/s; 13-point 3.94 vs 3.98 ms.  (Round 1's in-kernel band mapping of the dense grid was 6 % SLOWER:
            // it kept all XCDs on one x chunk with a rounded-up band size; here the list is simply permuted on the host.)
            std::vector<int32_t> out;
            out.reserve(cl.size());
            size_t i0 = 0;
            const int64_t per_chunk = (int64_t)tb_nyt * tb_nzt;
            while (i0 < cl.size()) {
               size_t i1 = i0;
               const int64_t xc = cl[i0] / per_chunk;
               while (i1 < cl.size() && cl[i1] / per_chunk == xc) i1++;
               const size_t n = i1 - i0, per = (n + 7) / 8;
               for (size_t p2 = 0; p2 < per; p2++)
                  for (size_t k = 0; k < 8; k++) { // balanced bands [k n / 8, (k+1) n / 8)
                     const size_t j0 = k * n / 8, j1 = (k + 1) * n / 8;
                     if (p2 < j1 - j0) out.push_back(cl[i0 + j0 + p2]);
                  }
               i0 = i1;
            }
            cl.swap(out);
         }
         tb_nclean = (int64_t)cl.size(); tb_ndirty = (int64_t)di.size(); tb_clean_cells = vol;
         int rc;
         if (tb_clean) { hipFree(tb_clean); tb_clean = nullptr; }
         if (tb_dirty) { hipFree(tb_dirty); tb_dirty = nullptr; }
         {
            // k_tb3 keeps u^{n+1} on the chip: the clean neighbours of a tile that steps singly (26-neighbourhood) are flagged to
            // store theirs, which that tile's second step reads
            std::vector<int32_t> clf(cl);
            if (triple && tb_ndirty > 0)
               for (auto &t : clf) {
                  const int zt = (int)(t % tb_nzt), yt = (int)((t / tb_nzt) % tb_nyt), xc = (int)(t / ((int64_t)tb_nzt * tb_nyt));
                  bool rim = false;
                  for (int a = std::max(xc - 1, 0); a <= std::min(xc + 1, tb_nxc - 1) && !rim; a++)
                     for (int b = std::max(yt - 1, 0); b <= std::min(yt + 1, tb_nyt - 1) && !rim; b++)
                        for (int c = std::max(zt - 1, 0); c <= std::min(zt + 1, tb_nzt - 1) && !rim; c++)
                           rim = dirty[((size_t)a * tb_nyt + b) * tb_nzt + c] != 0;
                  if (rim) t = (int32_t)((uint32_t)t | pf::TB3_RIM);
               }
            if ((rc = upload(&tb_clean, clf.data(), tb_nclean))) return rc;
         }
         if ((rc = upload(&tb_dirty, di.data(), tb_ndirty))) return rc;
         { // The placement search times the pair kernel on a SAMPLE of the clean tiles: every k-th x chunk, whole chunks in the
           // launch's own order (the effect it looks for is a property of how the four grids' pages lie relative to each
           // other, the same all along x), so a candidate costs 1/k of a launch (k = 4).
            int k = 4;
            if (tb_nxc < 8 * k) k = std::max(tb_nxc / 8, 1);
            std::vector<int32_t> sm;
            int64_t svol = 0;
            const int64_t per_chunk = (int64_t)tb_nyt * tb_nzt;
            for (int32_t t : cl) {
               const int xc = (int)(t / per_chunk);
               if (xc % k != k / 2) continue;
               sm.push_back(t);
               const int zt = (int)(t % tb_nzt), yt = (int)((t / tb_nzt) % tb_nyt);
               svol += (int64_t)(std::min(tbx0 + (xc + 1) * tb_chunk, tbx1) - (tbx0 + xc * tb_chunk)) *
                       (std::min(tby0 + (yt + 1) * TR, tby1) - (tby0 + yt * TR)) * (std::min(tbz0 + (zt + 1) * TC, tbz1) - (tbz0 + zt * TC));
            }
            if (tb_sample) { hipFree(tb_sample); tb_sample = nullptr; }
            tb_nsample = 0; tb_sample_frac = 1.0;
            if (k > 1 && !sm.empty() && vol > 0) {
               tb_nsample = (int64_t)sm.size(); tb_sample_frac = (double)svol / (double)vol;
               if ((rc = upload(&tb_sample, sm.data(), tb_nsample))) return rc;
            }
         }
         if (tb_nclean == 0) tb_xr.clear();
         if (fcc && !tb_xr.empty()) {
            // the single-step kernel's own tiling of the box's planes: 256 (fp64: 128) columns x 16 rows x the same x
            // chunks; a tile is needed unless every interior cell of it between the column strips lies in a clean core
            sh_nzt = (int)cdiv(P, 64 * V); sh_nyt = (int)cdiv(Ny - 2, 16);
            std::vector<int32_t> sh;
            for (int xc = 0; xc < tb_nxc; xc++)
               for (int yt = 0; yt < sh_nyt; yt++)
                  for (int zt = 0; zt < sh_nzt; zt++) {
                     const int ya = 1 + yt * 16, yb = std::min(ya + 16, (int)Ny - 1);
                     const int za = std::max(zt * 64 * V, tbz0), zb = std::min((zt + 1) * 64 * V, tbz1);
                     if (za >= zb) continue; // only column-strip cells: k_zstrip_fcc
                     bool need = ya < tby0 || yb > tby1;
                     // (Stepping the box's dirty tiles tile for tile, with the pair kernel's 12-row geometry, was measured in round 5 and
                     // LOSES: Musikverein, 28 % of the cells dirty, 4.11 ms per step against 3.80 with these 16-row tiles -- two rows per
                     // wave fetch twice the lines per cell.)
                     if (!need) {
                        const int t0y = (ya - tby0) / TR, t1y = (yb - 1 - tby0) / TR, t0z = (za - tbz0) / TC, t1z = (zb - 1 - tbz0) / TC;
                        for (int a = t0y; a <= t1y && !need; a++)
                           for (int c = t0z; c <= t1z && !need; c++) need = dirty[((size_t)xc * tb_nyt + a) * tb_nzt + c] != 0;
                     }
                     if (need) sh.push_back((int32_t)(((int64_t)xc * sh_nyt + yt) * sh_nzt + zt));
                  }
            sh_ntiles = (int64_t)sh.size();
            if (sh_tiles) { hipFree(sh_tiles); sh_tiles = nullptr; }
            if ((rc = upload(&sh_tiles, sh.data(), sh_ntiles))) return rc;
         }
      }
      if (triple && vol == 0) return PF_OK; // (every 20-row tile holds a receiver, a source or geometry: init_tb2 tries the pairs' smaller tiles)
      if (vbase == 40 && vol == 0) return set_err(PF_ERR_ARG, "air_variant 40 (temporal blocking) requested but the scene has no boundary-free tiles");
      // auto: the shell costs grow with the perimeter of the y-z cross-section, the gain with its area -- measured on
      // MI355X: 512^2 planes -4.5 %, 768^2 +9 %, 1024^2 +13 % for a box room; and two extra grids must be worth it.
      // Single-domain engines then time a blocked pair against the single-step kernels at creation (autotune()), so the
      // static rule only has to exclude the hopeless cases; slab engines have no such measurement and keep the strict one.
      if (vbase == 0 && single && ((double)vol < (fcc ? 0.5 : 0.35) * (double)(Nx * Ny * Nz) || tby1 - tby0 < 100 || tbz1 - tbz0 < 100)) return PF_OK;
      if (vbase == 0 && !single && ((double)vol < 0.6 * (double)(Nx * Ny * Nz) || tby1 - tby0 < 600 || tbz1 - tbz0 < 600)) return PF_OK;
      tb2_geom = true;
      if (!single) return PF_OK; // slab engines wait for pf_engine_set_spares (all four grids must be the caller's)
      int rc;
      bufC = try_dzalloc<Real>(npad);
      bufD = bufC ? try_dzalloc<Real>(npad) : nullptr;
      if (!bufD) { // no room for two more grids (state > ~45 % of the device memory): keep stepping singly
         if (bufC) hipFree(bufC);
         bufC = bufD = nullptr;
         tb2_geom = false;
         return PF_OK;
      }
      own_list.push_back(bufC); own_list.push_back(bufD);
      if (triple) {
         bufE = try_dzalloc<Real>(npad);
         if (!bufE) return PF_OK; // (no room for a fifth grid: init_tb2 falls back to pairs)
         own_list.push_back(bufE);
      }
      tb2 = true;
      { int rcw = init_walls(); if (rcw) return rcw; }
      if (triple) return PF_OK; // (with wall regions: triples; without: init_tb2 starts over with the pairs' geometry)
      if (wl_on) { zs_mode = 0; return PF_OK; } // (no single-step shell: nothing for the column-strip kernel to share)
      // Boundary nodes inside the column strips are updated by k_air_zstrip, which streams their lines anyway and holds
      // their six neighbours in registers (in k_boundary the floor / ceiling nodes of a box room -- stride-P neighbours,
      // one 128-byte line of u1 and of u0 per two nodes -- cost half of the pass: 0.30 of 0.63 ms at 1024^3).  The strip
      // kernel does the RIGID update only and leaves the result in u0b[li]; the branch ODEs of the lossy ones follow in
      // extra threads of the k_boundary launch, dense over the compact arrays (mode 2).  (Doing the ODEs inside the strip kernel as well was bit-identical
      // but slower -- they ran on the few lanes per wave that hold a node, 2.92 vs 2.59 ms per step -- and was retired.)
      // debug 0x20000000: mode 0, the list kernel visits every boundary node (the round-1 arrangement; also the fallback).
      zs_mode = fcc ? 0 : ((op.debug & 0x20000000) ? 0 : 2);
      if (Nb > 0 && !tb_xr.empty() && zs_mode == 2 && Nbl < ((int64_t)1 << 31)) {
         const int xb = tb_xr.front().first, xe = tb_xr.back().second;
         constexpr int V = pf::VecOf<Real>::V;
         const int nl = szl / V, nv = nl + (int)(P - szr) / V;
         std::vector<int64_t> hb(Nb);
         HIPCHK(hipMemcpy(hb.data(), d_bn, Nb * sizeof(int64_t), hipMemcpyDeviceToHost));
         std::vector<int32_t> hl(Nb), rest, sli, fd;
         std::vector<uint16_t> hadj(Nb), sadj;
         HIPCHK(hipMemcpy(hl.data(), d_lossy, Nb * sizeof(int32_t), hipMemcpyDeviceToHost));
         HIPCHK(hipMemcpy(hadj.data(), d_adj, Nb * sizeof(uint16_t), hipMemcpyDeviceToHost));
         // The boundary list is sorted by cell, so its strip nodes come in strip order (x, y, z): node k of the strips is the
         // k-th of them; a vector's record holds the number of its first node and one bit per cell with a node.
         std::vector<uint32_t> zv((size_t)(Nx * Ny * nv), 0u);
         rest.reserve(Nb);
         bool sorted = true;
         int64_t prev = -1;
         for (int64_t nb = 0; nb < Nb; nb++) {
            const int64_t ix = hb[nb] / plane, rem = hb[nb] % plane, iy = rem / P, iz = rem % P;
            const bool in_strip = ix >= xb && ix < xe && (iz < szl || iz >= szr);
            if (!in_strip) { rest.push_back((int32_t)nb); continue; }
            if (hb[nb] <= prev) { sorted = false; break; }
            prev = hb[nb];
            const int64_t v = iz < szl ? iz / V : nl + (iz - szr) / V, i = iz < szl ? iz % V : (iz - szr) % V;
            uint32_t &rec = zv[(size_t)((ix * Ny + iy) * nv + v)];
            if ((rec & 15u) == 0) rec = (uint32_t)sadj.size() << 4;
            rec |= 1u << i;
            sadj.push_back(hadj[nb]);
            sli.push_back(hl[nb]);
            if (hl[nb] >= 0) fd.push_back(hl[nb]);
         }
         if (!sorted || sadj.size() >= ((size_t)1 << 28)) zs_mode = 0; // (the engine sorts its lists: never seen)
         else {
            zs_nrest = (int64_t)rest.size();
            if ((rc = upload(&zs_map, zv.data(), (int64_t)zv.size()))) return rc;
            if ((rc = upload(&zs_adj, sadj.data(), (int64_t)sadj.size()))) return rc;
            if ((rc = upload(&zs_li, sli.data(), (int64_t)sli.size()))) return rc;
            if ((rc = upload(&zs_rest, rest.data(), zs_nrest))) return rc;
            zs_nfd = (int64_t)fd.size();
            if ((rc = upload(&zs_fd, fd.data(), zs_nfd))) return rc;
         }
      } else zs_mode = 0;
      return PF_OK;
   }

   // ---------------- wall regions: the shell of a blocked pair in pairs (pf_wall.h) ----------------
   // Six regions around the box: two whole-plane slabs normal to x, two row strips normal to y (the box's planes), two
   // column strips normal to z (the box's planes and rows) -- every interior cell outside the box belongs to exactly one.
   // Conditions: 7-point, margins that fit the pencils (8 cells for the strided ones,
   // 12 or 20 for the column strips), no source within one cell of the shell (a source is added BETWEEN the two steps,
   // which a region that keeps u^{n+1} in registers cannot see), fused boundary pass.  Otherwise the single-step shell of
   // round 2 runs (debug 0x10000000 forces that).
   // The frequency-dependent nodes are renumbered region by region in the order the lanes visit them (march, lane, pencil
   // cell), so that a wave's branch-state accesses are contiguous; the nodes inside the box follow in list order.
   // How a column strip becomes wall regions (0: it does not fit): 1 = one region with 12-cell pencils (strips of up to 10
   // columns), 3 = one region with 20-cell pencils (fp32), 2 = cut in two: the 7 columns next to the face, 12-cell pencils, and
   // the rest, plain air, 16-cell pencils.  t0 / t1: first / one past the last column of the box.
   int wl_lo_option(int t0, bool prefer_split) const {
      const bool split_ok = t0 >= 10 && t0 + 2 - 4 <= 16, wide_ok = sizeof(Real) == 4 && t0 + 2 <= 20;
      if (t0 + 2 <= 12) return 1;
      if (split_ok && (prefer_split || !wide_ok)) return 2;
      return wide_ok ? 3 : 0;
   }
   int wl_hi_option(int t1, bool prefer_split) const {
      const int z1 = std::min((t1 - 2) / 4 * 4, (int)P - 12), zw = (int)round_up(Nz - 12, 4), s0 = zw + 4, z2 = (t1 - 2) / 4 * 4; // (s0: whole vectors are stored)
      const int z20 = std::min((t1 - 2) / 4 * 4, (int)P - 20);
      const bool split_ok = zw >= 0 && zw + 12 <= P && s0 > t1 && s0 + 2 - z2 <= 16 && z2 + 16 <= P && (int)Nz - 1 - zw <= 11;
      const bool wide_ok = sizeof(Real) == 4 && z20 >= 0 && z20 + 20 >= Nz && t1 - z20 >= 2;
      if (z1 >= 0 && z1 + 12 >= Nz && t1 - z1 >= 2) return 1;
      if (split_ok && (prefer_split || !wide_ok)) return 2;
      return wide_ok ? 3 : 0;
   }
   void free_walls() {
      auto F = [](auto *&p) { if (p) hipFree((void *)p); p = nullptr; };
      F(wl_pen); F(wl_rec); F(wl_rest); F(wl_blk); F(vh1b); F(gh1b);
      wl_on = false;
      for (auto &g : wl_grp) g = WlGroup{};
   }
   // slab = true: a slab of a chain (pairs across two split-phase steps, step_begin): only the regions normal to y and z, over
   // the box's planes -- the planes between the slab's faces and the box (two edge planes per side, which are exchanged between
   // the two steps of a pair, and whatever x slab lies between them and the box) keep their single steps, and their boundary
   // nodes stay with the list kernel (wl_rest = the nodes of the interior planes no region owns).
   int init_walls(bool slab = false) {
      wl_on = false;
      const bool single = op.slab_first && op.slab_last;
      const bool vb = getenv("PFFDTD_VERBOSE") && atoi(getenv("PFFDTD_VERBOSE")) > 1;
#define WL_NO(why) do { if (vb) fprintf(stderr, "pffdtd_hip: no wall regions (%s)\n", why); return PF_OK; } while (0)
      if (slab ? (!tb2_slab || single) : (!tb2 || !single)) WL_NO("not a pair-stepping engine of this kind");
      if (fcc || tb_xr.empty() || swz || (op.debug & 0x10000000)) WL_NO("13-point / no box / exchanged axes / switched off");
      if (Nb > 0 && !fuse_boundary) WL_NO("boundary pass not fused");
      if (Nbl >= ((int64_t)1 << 24) || Nb >= ((int64_t)1 << 31)) WL_NO("too many nodes");
      constexpr int DPS = 8, V = pf::VecOf<Real>::V;
      if (!slab && (tbx0 + 2 > DPS || Nx - tbx1 + 2 > DPS)) WL_NO("x margins");
      if (tby0 + 2 > DPS || Ny - tby1 + 2 > DPS) WL_NO("y margins");
      if ((!slab && Nx < 2 * DPS) || Ny < 2 * DPS) WL_NO("grid too small");
      if (slab && (tbx0 < 3 || tbx1 > Nx - 3 || steps_done > 0)) WL_NO("slab: box reaches the edge planes, or stepping has begun"); // (the lossy arrays are re-ordered below: only before the first step)
      for (int64_t i = 0; i < Ns; i++) { // sources stay two cells inside the box
         int64_t ix, iy, iz;
         decode(sd.in_ixyz[i], ix, iy, iz);
         // (a slab's regions lie beside its box only: a source in the planes between a cut and the box is none of their business)
         if ((!slab && (ix < tbx0 + 1 || ix > tbx1 - 2)) || iy < tby0 + 1 || iy > tby1 - 2 || iz < tbz0 + 1 || iz > tbz1 - 2) WL_NO("a source within a cell of the shell");
      }
      std::vector<pf::WallRegion> reg;
      std::vector<int> dps, grp;
      auto mk = [&](int group, int mode, int nbase, int kg, int ko0, int ko1, int dp, int l0, int l1, int m0, int m1) {
         pf::WallRegion R{};
         R.mode = mode; R.nbase = nbase; R.kg = kg; R.ko0 = ko0; R.ko1 = ko1;
         R.kb0 = std::max(ko0 - 2, 0); R.kb1 = std::min(ko1 + 2, dp);
         R.l0 = l0; R.l1 = l1; R.m0 = m0; R.m1 = m1;
         // march steps per block.  Single domain (the regions run before the box kernel, the machine to themselves): 8 and 16
         // equal, 24-32 1-2 % slower, 48-64 7 % (1024^3).  Slab of a chain (few hundred blocks, beside the box kernel on a stream
         // of their own): a block is a chain of dependent steps, so short ones -- rank of 8: 0.473 (16) / 0.347 (8) / 0.334-0.354
         // (4) / 0.349 (3) / 0.358 (2) ms per step against 0.334-0.373 without the regions; rank of 4: 0.69 / 0.60 / 0.56-0.59
         // against 0.62-0.65; rank of 2: 1.06 (4-6) against 1.21; an end rank of 8: 0.338 against 0.388
         const int want = slab ? 4 : 16;
         const int len = m1 - m0, nmc = (int)std::max<int64_t>(cdiv(len, want), 1);
         R.mchunk = (int)cdiv(len, nmc);
         R.nlt = (int)cdiv(l1 - l0, pf::WALL_LT);
         R.nlp = R.nlt * pf::WALL_LT + 4;
         reg.push_back(R); dps.push_back(dp); grp.push_back(group);
      };
      if (!slab) {
         mk(0, 0, 0, 0, 1, tbx0, DPS, 1, (int)Nz - 1, 1, (int)Ny - 1);
         mk(0, 0, (int)Nx - DPS, DPS - 1, tbx1 - ((int)Nx - DPS), DPS - 1, DPS, 1, (int)Nz - 1, 1, (int)Ny - 1);
      }
      mk(0, 1, 0, 0, 1, tby0, DPS, 1, (int)Nz - 1, tbx0, tbx1);
      mk(0, 1, (int)Ny - DPS, DPS - 1, tby1 - ((int)Ny - DPS), DPS - 1, DPS, 1, (int)Nz - 1, tbx0, tbx1);
      // The column strips.  A strip of up to 10 columns is one region with pencils of 12 cells; a wider one (a sliver of the box went
      // to it) has pencils of 20 cells (fp32; ~350 registers, one wave per SIMD).  Cutting such a strip in two -- the 7 columns next
      // to the face with the wall layers, 12-cell pencils, and the rest, plain air, 16-cell pencils without any node code, two
      // waves per SIMD each -- was measured and LOSES (1024^3: 0.54 + 0.31 ms against 0.59 ms per pair; the strips' cost is the
      // 64 separate lines behind every load and store instruction, not the occupancy): only where the wide pencils do not fit (odd
      // widths, fp64) or with debug 0x2000000 (tests).
      int zb = 0;
      bool zok = true;
      const bool prefer_split = (op.debug & 0x2000000) != 0 || sizeof(Real) != 4;
      { // low side
         const int opt = wl_lo_option(tbz0, prefer_split);
         if (opt == 1) mk(1, 2, 0, 0, 1, tbz0, 12, tby0, tby1, tbx0, tbx1);
         else if (opt == 2) {
            mk(1, 2, 0, 0, 1, 8, 12, tby0, tby1, tbx0, tbx1);
            mk(2, 2, 4, -1, 8 - 4, tbz0 - 4, 16, tby0, tby1, tbx0, tbx1);
         } else if (opt == 3) mk(3, 2, 0, 0, 1, tbz0, 20, tby0, tby1, tbx0, tbx1);
         else zok = false;
      }
      { // high side
         const int z1 = std::min((tbz1 - 2) / 4 * 4, (int)P - 12);
         const int zw = (int)round_up(Nz - 12, 4), s0 = zw + 4;   // wall part: pencil from column zw, owned from s0 (whole vectors are stored)
         const int z2 = (tbz1 - 2) / 4 * 4;                      // the rest: pencil from column z2
         const int z20 = std::min((tbz1 - 2) / 4 * 4, (int)P - 20);
         const int opt = wl_hi_option(tbz1, prefer_split);
         if (opt == 1) { zb = z1; mk(1, 2, z1, (int)Nz - 1 - z1, tbz1 - z1, (int)Nz - 1 - z1, 12, tby0, tby1, tbx0, tbx1); }
         else if (opt == 2) {
            zb = zw;
            mk(1, 2, zw, (int)Nz - 1 - zw, 4, (int)Nz - 1 - zw, 12, tby0, tby1, tbx0, tbx1);
            mk(2, 2, z2, -1, tbz1 - z2, s0 - z2, 16, tby0, tby1, tbx0, tbx1);
         } else if (opt == 3) { zb = z20; mk(3, 2, z20, (int)Nz - 1 - z20, tbz1 - z20, (int)Nz - 1 - z20, 20, tby0, tby1, tbx0, tbx1); }
         else zok = false;
      }
      if (!zok) WL_NO("column strips too wide for the pencils");
#undef WL_NO
      const int nregs = (int)reg.size();
      // every feasibility check comes BEFORE a device array is touched (the lossy arrays are re-ordered in place below)
      std::vector<uint32_t> rloc((size_t)nregs); // a region's place in its launch group
      {
         int cnt[4] = {0, 0, 0, 0};
         for (int i = 0; i < nregs; i++) {
            if (cnt[grp[i]] >= pf::WALL_MAXREG) return PF_OK;
            rloc[i] = (uint32_t)cnt[grp[i]]++;
         }
      }
      int64_t npen = 0;
      for (int i = 0; i < nregs; i++) { reg[i].pen_off = npen; npen += (int64_t)(reg[i].m1 - reg[i].m0 + 2) * reg[i].nlp; }
      if (npen >= ((int64_t)1 << 31)) return PF_OK;
      // a node's place in a region's frame
      auto frame = [&](const pf::WallRegion &R, int64_t ix, int64_t iy, int64_t iz, int &k, int &lc, int &m) {
         if (R.mode == 0) { k = (int)ix - R.nbase; lc = (int)iz; m = (int)iy; }
         else if (R.mode == 1) { k = (int)iy - R.nbase; lc = (int)iz; m = (int)ix; }
         else { k = (int)iz - R.nbase; lc = (int)iy; m = (int)ix; }
      };
      std::vector<int64_t> hb(Nb);
      std::vector<uint16_t> hadj(Nb);
      std::vector<int32_t> hl(Nb, -1);
      if (Nb) {
         HIPCHK(hipMemcpy(hb.data(), d_bn, Nb * sizeof(int64_t), hipMemcpyDeviceToHost));
         HIPCHK(hipMemcpy(hadj.data(), d_adj, Nb * sizeof(uint16_t), hipMemcpyDeviceToHost));
         HIPCHK(hipMemcpy(hl.data(), d_lossy, Nb * sizeof(int32_t), hipMemcpyDeviceToHost));
      }
      // owners, and the new order of the frequency-dependent nodes
      std::vector<int8_t> owner(Nb, 8);
      struct Key { int32_t r, m, lc, k, li; };
      std::vector<Key> keys;
      keys.reserve((size_t)Nbl);
      std::vector<int32_t> rest;
      for (int64_t nb = 0; nb < Nb; nb++) {
         const int64_t ix = hb[nb] / plane, rem = hb[nb] % plane, iy = rem / P, iz = rem % P;
         int r = 8, k = 0, lc = 0, m = 0;
         for (int i = 0; i < nregs; i++) {
            int kk, ll, mm;
            frame(reg[i], ix, iy, iz, kk, ll, mm);
            if (kk >= reg[i].ko0 && kk < reg[i].ko1 && ll >= reg[i].l0 && ll < reg[i].l1 && mm >= reg[i].m0 && mm < reg[i].m1) { r = i; k = kk; lc = ll; m = mm; break; }
         }
         owner[nb] = (int8_t)r;
         const Range &midr = tb3_geom ? bn_mid3 : bn_mid2;
         if (r == 8 && (!slab || (nb >= midr.b && nb < midr.e))) rest.push_back((int32_t)nb); // (a slab's edge planes: range launches)
         if (hl[nb] >= 0) keys.push_back({r, m, lc, k, hl[nb]});
      }
      if ((int64_t)keys.size() != Nbl) return PF_OK; // (a lossy node that is no boundary node: fuse_boundary excludes it)
      std::stable_sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
         if (a.r != b.r) return a.r < b.r;
         if (a.r == 8) return a.li < b.li;
         if (a.m != b.m) return a.m < b.m;
         if (a.lc != b.lc) return a.lc < b.lc;
         return a.k < b.k;
      });
      std::vector<int32_t> newli((size_t)Nbl);
      for (int64_t j = 0; j < Nbl; j++) newli[keys[j].li] = (int32_t)j;
      // pencil tables: node masks, then the records in pencil order (the boundary list is sorted by cell, so a pencil's
      // nodes arrive in ascending pencil-cell order)
      std::vector<uint4> pen((size_t)npen, make_uint4(0u, 0u, 0u, 0u));
      auto visit = [&](auto &&fn) {
         for (int64_t nb = 0; nb < Nb; nb++) {
            const int64_t ix = hb[nb] / plane, rem = hb[nb] % plane, iy = rem / P, iz = rem % P;
            for (int i = 0; i < nregs; i++) {
               const pf::WallRegion &R = reg[i];
               int k, lc, m;
               frame(R, ix, iy, iz, k, lc, m);
               if (k < std::max(1, R.ko0 - 1) || k > std::min(dps[i] - 2, R.ko1) || lc < R.l0 - 1 || lc > R.l1 || m < R.m0 - 1 || m > R.m1) continue;
               fn(nb, R.pen_off + (int64_t)(m - (R.m0 - 1)) * R.nlp + (lc - (R.l0 - 2)), k);
            }
         }
      };
      visit([&](int64_t, int64_t pi, int k) { pen[(size_t)pi].x |= 1u << k; });
      int64_t nrec = 0;
      for (int64_t i = 0; i < npen; i++) { pen[(size_t)i].y = (uint32_t)nrec; nrec += __builtin_popcount(pen[(size_t)i].x); }
      if (nrec >= ((int64_t)1 << 27)) return PF_OK;
      std::vector<uint32_t> rec((size_t)std::max<int64_t>(nrec, 1), 0u);
      visit([&](int64_t nb, int64_t pi, int k) {
         uint4 &e = pen[(size_t)pi];
         const uint32_t jn = (uint32_t)__builtin_popcount(e.x & ((1u << k) - 1u));
         const uint32_t adj = (uint32_t)(hadj[nb] & 63u);
         rec[(e.y & 0x7ffffffu) + jn] = adj | (hl[nb] >= 0 ? (0x40u | ((uint32_t)newli[hl[nb]] << 8)) : 0u);
         if (jn < 5) { // the entry itself carries the first five nodes: adjacency bits, lossy flags, the first lossy one's place
            e.z |= adj << (6 * jn);
            if (hl[nb] >= 0) {
               if ((e.w & 31u) == 0u) { e.w |= (uint32_t)newli[hl[nb]] << 8; e.y |= (uint32_t)k << 27; }
               e.w |= 1u << jn;
            }
         }
      });
      int rc;
      free_walls(); // (a second call -- set_spares after place_grids -- must not leak the first one's tables)
      // Wall regions are an optimisation: when the device has no room for their tables and the second copy of the branch state, the
      // engine keeps the single-step shell instead of failing (everything is allocated BEFORE the lossy arrays are re-ordered).
      auto no_room = [&]() { free_walls(); (void)hipGetLastError(); g_err.clear(); if (vb) fprintf(stderr, "pffdtd_hip: no wall regions (no device memory for their tables)\n"); return PF_OK; };
      vh1b = try_dzalloc<Real>(round_up(Nbl, 64) * PF_MMB);
      gh1b = vh1b ? try_dzalloc<Real>(round_up(Nbl, 64) * PF_MMB) : nullptr;
      if (!gh1b) return no_room();
      if ((rc = upload(&wl_pen, pen.data(), npen))) return no_room();
      if ((rc = upload(&wl_rec, rec.data(), nrec))) return no_room();
      wl_nrest = (int64_t)rest.size();
      if ((rc = upload(&wl_rest, rest.data(), wl_nrest))) return no_room();
      // the lossy arrays in the new order (state and node-value arrays are all zeros at creation)
      if (Nbl) {
         std::vector<int64_t> bl(Nbl), bl2(Nbl);
         std::vector<Real> sf(Nbl), sf2(Nbl);
         std::vector<int8_t> mt(Nbl), mt2(Nbl);
         HIPCHK(hipMemcpy(bl.data(), d_bnl, Nbl * sizeof(int64_t), hipMemcpyDeviceToHost));
         HIPCHK(hipMemcpy(sf.data(), d_ssaf, Nbl * sizeof(Real), hipMemcpyDeviceToHost));
         HIPCHK(hipMemcpy(mt.data(), d_mat, Nbl * sizeof(int8_t), hipMemcpyDeviceToHost));
         for (int64_t j = 0; j < Nbl; j++) { bl2[newli[j]] = bl[j]; sf2[newli[j]] = sf[j]; mt2[newli[j]] = mt[j]; }
         HIPCHK(hipMemcpy(d_bnl, bl2.data(), Nbl * sizeof(int64_t), hipMemcpyHostToDevice));
         HIPCHK(hipMemcpy(d_ssaf, sf2.data(), Nbl * sizeof(Real), hipMemcpyHostToDevice));
         HIPCHK(hipMemcpy(d_mat, mt2.data(), Nbl * sizeof(int8_t), hipMemcpyHostToDevice));
         for (int64_t nb = 0; nb < Nb; nb++) if (hl[nb] >= 0) hl[nb] = newli[hl[nb]];
         HIPCHK(hipMemcpy(d_lossy, hl.data(), Nb * sizeof(int32_t), hipMemcpyHostToDevice));
      }
      for (auto &g : wl_grp) g = WlGroup{};
      for (int i = 0; i < nregs; i++) {
         WlGroup &g = wl_grp[grp[i]];
         g.reg[g.nreg++] = reg[i];
      }
      // Block lists.  A block (lane tile x march chunk of a region) whose pencils all have the same structure and that touches no
      // ghost / ABC cell along its lane and march axes goes to the FAST launch with that structure attached; the others
      // (edges, corners, the ends of a march) to the generic one.
      std::vector<uint4> lists[12]; // group g: 3 g alike, 3 g + 1 generic, 3 g + 2 alike and free of nodes
      int64_t nfast = 0, ngen = 0;
      for (int i = 0; i < nregs; i++) {
         const pf::WallRegion &R = reg[i];
         const int NL = R.mode == 2 ? (int)Ny : (int)Nz, NM = R.mode == 0 ? (int)Ny : (int)Nx;
         const int nmc = (int)cdiv(R.m1 - R.m0, R.mchunk);
         for (int c = 0; c < nmc; c++)
            for (int jt = 0; jt < R.nlt; jt++) {
               const int ms = R.m0 + c * R.mchunk, me = std::min(ms + R.mchunk, R.m1);
               const int lc0 = R.l0 - 1 + pf::WALL_LT * jt, lc1 = std::min(R.l0 + pf::WALL_LT * jt + pf::WALL_LT, R.l1); // lanes 1 .. 62 within the region
               bool fast = lc0 >= 2 && lc1 <= NL - 3 && ms - 1 >= 2 && me <= NM - 3 && !(op.debug & 0x8000000);
               const uint4 ref = pen[(size_t)(R.pen_off + (int64_t)(ms - 1 - (R.m0 - 1)) * R.nlp + (lc0 - (R.l0 - 2)))];
               if (__builtin_popcount(ref.x) > 5 || __builtin_popcount(ref.w & 31u) > 1) fast = false;
               for (int m = ms - 1; m <= me && fast; m++) {
                  const uint4 *row = pen.data() + (size_t)(R.pen_off + (int64_t)(m - (R.m0 - 1)) * R.nlp - (R.l0 - 2));
                  for (int lc = lc0; lc <= lc1; lc++) {
                     const uint4 &e = row[lc];
                     if (e.x != ref.x || e.z != ref.z || (e.w & 31u) != (ref.w & 31u) || (e.y >> 27) != (ref.y >> 27)) { fast = false; break; }
                  }
               }
               const uint32_t rl = rloc[i];
               const uint4 b = make_uint4(rl | ((uint32_t)jt << 3) | ((uint32_t)c << 16), ref.x, ref.z, (ref.w & 31u) | ((ref.y >> 27) << 8));
               lists[3 * grp[i] + (fast ? (ref.x == 0u && R.mode == 2 ? 2 : 0) : 1)].push_back(b);
               (fast ? nfast : ngen)++;
            }
      }
      {
         std::vector<uint4> all;
         for (int q = 0; q < 12; q++) { wl_grp[q / 3].blk0[q % 3] = (uint32_t)all.size(); wl_grp[q / 3].nblk[q % 3] = (uint32_t)lists[q].size(); all.insert(all.end(), lists[q].begin(), lists[q].end()); }
         if ((rc = upload(&wl_blk, all.data(), (int64_t)all.size()))) return no_room(); // (the lossy arrays are re-ordered by now: consistently, which any path accepts)
      }
      wl_on = true;
      if (getenv("PFFDTD_VERBOSE") && atoi(getenv("PFFDTD_VERBOSE")) > 0)
         fprintf(stderr, "pffdtd_hip: wall regions: box x [%d,%d) y [%d,%d) z [%d,%d), %ld pencils, %ld node records, %ld of %ld boundary nodes left to the list kernel, column pencils of %d cells from column %d; %ld blocks alike, %ld generic\n",
                 tbx0, tbx1, tby0, tby1, tbz0, tbz1, (long)npen, (long)nrec, (long)wl_nrest, (long)Nb, dps.back(), zb, (long)nfast, (long)ngen);
      return PF_OK;
   }
   // both steps of the wall regions: A = u^{n-1}, B = u^n -> C = u^{n+1}, D = u^{n+2}; branch state vh1 / gh1 -> vh1b / gh1b;
   // node values: P2 = u^{n-1} and P1 = u^n are read, P0 <- u^{n+1}, P1 <- u^{n+2}
   // (s_gen: the stream of the generic blocks -- edges, corners: few waves, each a long chain of dependent steps)
   void launch_walls(hipStream_t s, hipStream_t s_gen, const Real *A, const Real *B, Real *C, Real *D, Real *P0, Real *P1, const Real *P2) {
      pf::WallParams<Real> wp{};
      wp.A = A; wp.B = B; wp.C = C; wp.D = D;
      wp.plane = plane; wp.Nx = (int)Nx; wp.Ny = (int)Ny; wp.Nz = (int)Nz; wp.P = (int)P; wp.first = op.slab_first; wp.last = op.slab_last;
      wp.pen = wl_pen; wp.rec = wl_rec;
      wp.sv_in = vh1; wp.sg_in = gh1; wp.sv_out = vh1b; wp.sg_out = gh1b;
      wp.x2 = P2; wp.x1 = P1; wp.o1 = P0; wp.o2 = P1;
      wp.ssaf = d_ssaf; wp.mat = d_mat; wp.Mb = d_Mb; wp.mq = d_mq; wp.beta = d_beta;
      wp.lo2 = lo2; wp.sl2 = sl2; wp.l = l; wp.mmax = mb_max; wp.nmat = sd.Nm;
      for (int gi = 0; gi < 4; gi++) {
         const WlGroup &g = wl_grp[gi];
         wp.nreg = g.nreg;
         for (int i = 0; i < g.nreg; i++) wp.reg[i] = g.reg[i];
         for (int q = 0; q < 3; q++) { // alike blocks, generic blocks, alike blocks without nodes (column strips only)
            if (!g.nblk[q]) continue;
            wp.blk = wl_blk + g.blk0[q];
            const dim3 gd(g.nblk[q]), b(64);
            hipStream_t st = q == 1 ? s_gen : s;
#define PF_WALL_N(DP, VEC, S) do { if (q == 2) { if constexpr (VEC) hipLaunchKernelGGL((pf::k_wall2<Real, DP, VEC, true, false, 12, S>), gd, b, 0, st, wp, a1, a2); } \
                              else if (mb_max <= 4) { \
                                 if (q == 0) hipLaunchKernelGGL((pf::k_wall2<Real, DP, VEC, true, true, 4, S>), gd, b, 0, st, wp, a1, a2); \
                                 else hipLaunchKernelGGL((pf::k_wall2<Real, DP, VEC, false, true, 4, S>), gd, b, 0, st, wp, a1, a2); \
                              } else { \
                                 if (q == 0) hipLaunchKernelGGL((pf::k_wall2<Real, DP, VEC, true, true, 12, S>), gd, b, 0, st, wp, a1, a2); \
                                 else hipLaunchKernelGGL((pf::k_wall2<Real, DP, VEC, false, true, 12, S>), gd, b, 0, st, wp, a1, a2); } } while (0)
#define PF_WALL(DP, VEC) do { if (sg) PF_WALL_N(DP, VEC, true); else PF_WALL_N(DP, VEC, false); } while (0)
            if (gi == 0) PF_WALL(8, false);
            else if (gi == 1) PF_WALL(12, true);
            else if (gi == 2) PF_WALL(16, true);
            else if constexpr (sizeof(Real) == 4) PF_WALL(20, true);
#undef PF_WALL
#undef PF_WALL_N
         }
      }
   }
   // steps n and n+1 with the shell in pairs as well.  Order: box (both steps), then the first step of what no wall region
   // owns -- the box's dirty tiles and the boundary nodes inside it --, source / receivers of step n, the wall regions (both
   // steps; they read u^{n-1}, u^n and the old branch state only), then the second step of the dirty tiles and their nodes.
   int step_pair_walls(int64_t n) {
      hipStream_t s = s_main;
      Real *A = u0, *B = u1, *C = bufC, *D = bufD;
      Real *P0 = ub[0], *P1 = ub[1], *P2 = ub[2];
      auto get_ev = [&]() { std::pair<hipEvent_t, hipEvent_t> e{}; if (!ev_pool.empty()) { e = ev_pool.back(); ev_pool.pop_back(); } else { hipEventCreate(&e.first); hipEventCreate(&e.second); } return e; };
      std::pair<hipEvent_t, hipEvent_t> ev{}, ev2{}, evt{}, eva{};
      // "air" of a pair with wall regions (pf_timing.air_ms_total, the CLI's "Air update" line): the alike blocks' launches and the
      // box kernel on the main stream -- the regions' boundary nodes are inside those launches and cannot be told apart
      if (op.timing) { ev = get_ev(); ev2 = get_ev(); evt = get_ev(); eva = get_ev(); hipEventRecord(ev.first, s); hipEventRecord(eva.first, s); }
      // The wall regions read u^{n-1}, u^n and the old branch state only and write cells the box kernel does not: any order will do.
      // The generic blocks (edges, corners: a few hundred waves, each a long chain of dependent steps) go to the second stream
      // and run beside the alike blocks' launches, which are issue-bound; beside the bandwidth-bound box kernel they crawl
      // (measured: 0.47 -> 3.6 ms), so that one comes after.  debug 0x4000000: everything on the main stream.
      const bool beside = !(op.debug & 0x4000000);
      hipStream_t sw = beside ? s_edge : s_main;
      if (beside) { HIPCHK(hipEventRecord(ev_pre, s_main)); HIPCHK(hipStreamWaitEvent(s_edge, ev_pre, 0)); }
      u0_src = A; u1 = B; u0 = C;
      launch_dirty_tiles(sw);
      bnd_sel = wl_rest; bs_vout = vh1b; bs_gout = gh1b;
      launch_rigid(sw, {0, wl_nrest});
      launch_walls(s, sw, A, B, C, D, P0, P1, P2); // (every wall launch beside the box kernel instead of before it: 492 vs 511-516 Gvox/s)
      if (op.timing) hipEventRecord(evt.first, s);
      launch_tb2(s, A, B, C, D);
      if (op.timing) { hipEventRecord(evt.second, s); tb2_ev.push_back(evt); hipEventRecord(eva.second, s); air_ev.push_back(eva); }
      if (beside) { HIPCHK(hipEventRecord(ev_edge, s_edge)); HIPCHK(hipStreamWaitEvent(s_main, ev_edge, 0)); }
      launch_io(s, n, true, {0, Ns}); // (receivers read u^n; the source goes into u^{n+1}, which only the second step below reads)
      if (ring_fill == 0) ring_n0 = n;
      ring_fill++; steps_done++;
      if (op.timing) { hipEventRecord(ev.second, s); step_ev.push_back(ev); hipEventRecord(ev2.first, s); }
      std::swap(vh1, vh1b); std::swap(gh1, gh1b); // the state after the pair (the nodes inside the box: after its first step)
      u0_src = B; u1 = C; u0 = D;
      launch_dirty_tiles(s);
      bs_vout = bs_gout = nullptr;
      ub[0] = P1; ub[2] = P1; // second step of the box's nodes: u2b = u^n of the node, overwritten by its u^{n+2} (where the regions put theirs)
      launch_rigid(s, {0, wl_nrest});
      ub[0] = P2; ub[1] = P1; ub[2] = P0;
      bnd_sel = nullptr;
      launch_io(s, n + 1, true, {0, Ns});
      ring_fill++; steps_done++;
      u0_src = nullptr; u0 = C; u1 = D; bufC = A; bufD = B;
      if (tb3) tb3_pick();
      if (op.timing) { hipEventRecord(ev2.second, s); step_ev.push_back(ev2); }
      HIPCHK(hipGetLastError());
      if (ring_fill == ring_depth) return flush();
      return PF_OK;
   }
   // tb3: which grids does the next blocked step write?  Where the five grids lie relative to each other decides the speed of k_tb3
   // (grid placement): the assignment measured at creation -- state home[0], home[1] -> home[2], home[3] and back, home[4]
   // the u^{n+1} grid -- is kept wherever the state allows; after a pair or an odd number of single steps (the end of a run) the
   // next triple is one step off that cycle and returns to it.
   void tb3_remember_home() { home[0] = u0; home[1] = u1; home[2] = bufD; home[3] = bufE; home[4] = bufC; }
   void tb3_pick() {
      if (!home[0]) return;
      if (u0 == home[0] && u1 == home[1]) { bufD = home[2]; bufE = home[3]; bufC = home[4]; return; }
      if (u0 == home[2] && u1 == home[3]) { bufD = home[0]; bufE = home[1]; bufC = home[4]; return; }
      Real *fr[3];
      int nf = 0;
      for (Real *g : home) if (g != u0 && g != u1 && nf < 3) fr[nf++] = g;
      if (nf != 3) return; // (cannot happen: the state is two of the five)
      auto is_free = [&](Real *g) { return g == fr[0] || g == fr[1] || g == fr[2]; };
      if (is_free(home[0]) && is_free(home[1])) { bufD = home[0]; bufE = home[1]; }
      else if (is_free(home[2]) && is_free(home[3])) { bufD = home[2]; bufE = home[3]; }
      else { bufD = fr[0]; bufE = fr[1]; }
      for (Real *g : fr) if (g != bufD && g != bufE) bufC = g;
   }
   // steps n, n+1 and n+2 in one go (tb3): the box by k_tb3 (u^{n+1} stays on the chip), the shell's first two steps as wall regions
   // (k_wall2, exactly as in step_pair_walls), its third as a single step out of memory (the lean kernel on the x slabs and row
   // strips, k_air_zstrip on the column strips, k_boundary over every node); the tiles that step singly (sources, receivers,
   // geometry inside the box) and the box's own boundary nodes take three single steps, the second of which reads the u^{n+1}
   // their flagged neighbours left in bufC.  State (u0, u1) -> (bufD, bufE); the old state grids become the next triple's targets.
   int step_triple(int64_t n) {
      if (n < 0 || n + 2 >= Nt) return set_err(PF_ERR_ARG, "step triple %ld outside [0,Nt=%ld)", (long)n, (long)Nt);
      hipStream_t s = s_main;
      Real *A = u0, *B = u1, *C = bufC, *D = bufD, *E = bufE;
      Real *P0 = ub[0], *P1 = ub[1], *P2 = ub[2]; // node values: P2 = u^{n-1}, P1 = u^n, P0 free
      auto get_ev = [&]() { std::pair<hipEvent_t, hipEvent_t> e{}; if (!ev_pool.empty()) { e = ev_pool.back(); ev_pool.pop_back(); } else { hipEventCreate(&e.first); hipEventCreate(&e.second); } return e; };
      std::pair<hipEvent_t, hipEvent_t> ev{}, evt{}, eva{};
      if (op.timing) { ev = get_ev(); evt = get_ev(); eva = get_ev(); hipEventRecord(ev.first, s); hipEventRecord(eva.first, s); }
      const bool beside = !(op.debug & 0x4000000);
      hipStream_t sw = beside ? s_edge : s_main; // the generic wall blocks and the first step of the single-step tiles: beside the alike blocks
      if (beside) { HIPCHK(hipEventRecord(ev_pre, s_main)); HIPCHK(hipStreamWaitEvent(s_edge, ev_pre, 0)); }
      // ---- step n: single-step tiles and the box's own nodes A, B -> C; wall regions A, B -> C, D; box A, B -> D, E
      u0_src = A; u1 = B; u0 = C;
      launch_dirty_tiles(sw);
      bnd_sel = wl_rest; bs_vout = vh1b; bs_gout = gh1b;
      launch_rigid(sw, {0, wl_nrest});
      // (debug 0x10000, an experiment: the alike blocks too beside k_tb3 instead of before it)
      launch_walls((beside && (op.debug & 0x10000)) ? s_edge : s, sw, A, B, C, D, P0, P1, P2);
      if (op.timing) hipEventRecord(evt.first, s);
      launch_tb3(s, A, B, C, D, E);
      if (op.timing) { hipEventRecord(evt.second, s); tb2_ev.push_back(evt); hipEventRecord(eva.second, s); air_ev.push_back(eva); }
      if (beside) { HIPCHK(hipEventRecord(ev_edge, s_edge)); HIPCHK(hipStreamWaitEvent(s_main, ev_edge, 0)); }
      launch_io(s, n, true, {0, Ns}); // receivers read u^n (B); the source goes into u^{n+1} (C), which only the single-step tiles read
      if (ring_fill == 0) ring_n0 = n;
      ring_fill++; steps_done++;
      if (op.timing) { hipEventRecord(ev.second, s); step_ev.push_back(ev); ev = get_ev(); hipEventRecord(ev.first, s); }
      std::swap(vh1, vh1b); std::swap(gh1, gh1b); // the state after two steps (the nodes inside the box: after their first)
      // ---- step n+1: single-step tiles and their nodes B, C -> D (the regions and the box have theirs)
      u0_src = B; u1 = C; u0 = D;
      launch_dirty_tiles(s);
      bs_vout = bs_gout = nullptr;
      ub[0] = P1; ub[2] = P1; // u2b = u^n of the node, overwritten by its u^{n+2} (where the regions put theirs)
      launch_rigid(s, {0, wl_nrest});
      bnd_sel = nullptr;
      launch_io(s, n + 1, true, {0, Ns}); // receivers read u^{n+1} (C: shell and single-step tiles hold it); source into u^{n+2} (D)
      ring_fill++; steps_done++;
      if (op.timing) { hipEventRecord(ev.second, s); step_ev.push_back(ev); ev = get_ev(); hipEventRecord(ev.first, s); eva = get_ev(); hipEventRecord(eva.first, s); }
      // ---- step n+2: the whole shell and the single-step tiles C, D -> E as ONE single step, every boundary node by the list kernel
      // (node values: u^{n+1} in P0, u^{n+2} in P1 -> u^{n+3} into P2; branch state in place)
      u0_src = C; u1 = D; u0 = E;
      ub[0] = P2; ub[1] = P1; ub[2] = P0;
      launch_shell(s);
      if (op.timing) { hipEventRecord(eva.second, s); air_ev.push_back(eva); }
      launch_rigid(s, {0, Nb});
      launch_fd(s, {0, Nbl});
      launch_io(s, n + 2, true, {0, Ns});
      ring_fill++; steps_done++;
      ub[0] = P0; ub[1] = P2; ub[2] = P1; // (newest in ub[1], the one before in ub[2], ub[0] free: the single steps' convention)
      u0_src = nullptr; u0 = D; u1 = E; bufD = A; bufE = B; // bufC stays the u^{n+1} grid ...
      tb3_pick();                                           // ... on the placed cycle; off it: back towards it
      if (op.timing) { hipEventRecord(ev.second, s); step_ev.push_back(ev); }
      HIPCHK(hipGetLastError());
      if (ring_fill == ring_depth) return flush();
      return PF_OK;
   }
   // slab engines are created with the triples' box and tiles where those exist (init_tb2); a caller that hands over four grids
   // only, or whose wall regions do not fit that box, gets the pairs' geometry instead (before the first step)
   int pairs_geometry() {
      if (!tb3_geom) return PF_OK;
      free_walls();
      tb3_geom = tb3_slab = false;
      return init_tb2_impl(false);
   }
   int set_spares(void *g2, void *g3) override {
      if (in_step || pair_phase) return set_err(PF_ERR_STATE, "pf_engine_set_spares inside a step");
      if (!g2 || !g3) return set_err(PF_ERR_ARG, "pf_engine_set_spares: null grid");
      if (steps_done == 0) { int rcg = pairs_geometry(); if (rcg) return rcg; }
      if (!tb2_geom || (op.slab_first && op.slab_last)) return 1; // not an error: this engine keeps stepping singly
      bufC = (Real *)g2; bufD = (Real *)g3;
      tb2_slab = true;
      if (!wl_on && !(op.debug & 0x10000000)) { int rcw = init_walls(true); if (rcw) return rcw; }
      return PF_OK;
   }
   // ---- which interior path?  Measured, not guessed: at creation the candidates run three times each on the real grids,
   // writing to scratch (the state is not touched): lean fused kernel, barrier-free kernel with virtual ghosts, and --
   // where a box exists -- a temporally blocked pair incl. its shell.  The boundary pass and the I/O are common to all.
   // (7-point only; explicit air_variant requests and debug 0x8000 skip it.)  Sizes decide in ways no static rule
   // caught: 1024^3 fp32 pair 411 > lean 377 > barrier-free 364 Gvox/s, 896^3 barrier-free 362 > pair 335 > lean 303.
   // 13-point: one in-place-equivalent single step (written to scratch) against half a blocked pair with its shell
   // 13-point single steps: lanes per row segment of k_air_fcc (64 / 32 / 16) measured where they pad the rows differently --
   // the static rule asks for a 25 % narrower padded row before it leaves 64 lanes, which rooms stored along their longest axis
   // (2852 columns = 11.1 segments of 256) never offer, although the half-empty last segment costs them 7 % of the lanes
   int autotune_fcc_lw() {
      if (!fcc || !abck || sg || vbase != 0 || (op.debug & 0x8300) || lw_force || !(op.slab_first && op.slab_last)) return PF_OK;
      if (Nx * Ny * Nz < ((int64_t)1 << 22)) return PF_OK;
      constexpr int V = pf::VecOf<Real>::V;
      Real *scr = try_dzalloc<Real>(npad);
      if (!scr) return PF_OK;
      hipEvent_t e0, e1;
      HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
      Real *U0 = u0;
      u0_src = U0; u0 = scr; // (out of place: the state is not touched)
      int best_lw = 0;
      float best = 0;
      int64_t seen = -1;
      for (int lw : {64, 32, 16}) {
         const int64_t w = cdiv(P, (int64_t)lw * V) * lw * V;
         if (w == seen) continue; // same padded width as the wider segment: the wider one wins anyway
         seen = w;
         lw_force = lw;
         launch_air_march(s_main, 1, (int)Nx - 1);
         hipEventRecord(e0, s_main);
         for (int i = 0; i < 3; i++) launch_air_march(s_main, 1, (int)Nx - 1);
         hipEventRecord(e1, s_main);
         hipEventSynchronize(e1);
         float ms = 0;
         hipEventElapsedTime(&ms, e0, e1);
         if (best_lw == 0 || ms < 0.98f * best) { best = ms; best_lw = lw; }
      }
      lw_force = best_lw;
      // ... and the tile order: XCD-banded (the rule for large planes) against the plain order.  Rooms stored along their longest
      // axis have long rows (Musikverein: 23 segments of 128 columns) and run 1-3 % faster, and steadier, in the plain order
      // (345.0-345.3 against 335-342 Gvox/s in alternating runs); cubes keep the banded one (2.32 against 2.46 ms at 1024^3).
      for (int mode : {0}) {
         order_force = mode;
         launch_air_march(s_main, 1, (int)Nx - 1);
         hipEventRecord(e0, s_main);
         for (int i = 0; i < 3; i++) launch_air_march(s_main, 1, (int)Nx - 1);
         hipEventRecord(e1, s_main);
         hipEventSynchronize(e1);
         float ms = 0;
         hipEventElapsedTime(&ms, e0, e1);
         if (ms < 0.985f * best) best = ms; else order_force = -1;
      }
      tune_ms[1] = best / 3;
      u0 = U0; u0_src = nullptr;
      hipEventDestroy(e0); hipEventDestroy(e1);
      HIPCHK(hipStreamSynchronize(s_main));
      hipFree(scr);
      return hipGetLastError() == hipSuccess ? PF_OK : set_err(PF_ERR_HIP, "13-point segment-width measurement: kernel launch failed");
   }
   int autotune_fcc() {
      if (!tb2) return autotune_fcc_lw();
      if (vbase != 0 || (op.debug & 0x8000)) return PF_OK;
      hipEvent_t e0, e1;
      HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
      HIPCHK(hipDeviceSynchronize());
      auto timed = [&](auto &&fn) -> float {
         fn();
         hipEventRecord(e0, s_main);
         for (int i = 0; i < 3; i++) fn();
         hipEventRecord(e1, s_main);
         hipEventSynchronize(e1);
         float ms = 0;
         hipEventElapsedTime(&ms, e0, e1);
         return ms / 3;
      };
      Real *U0 = u0, *U1 = u1;
      u0_src = U0; u0 = bufC;
      for (int i = 0; i < 8 && (double)i * (double)(Nx * Ny * Nz) < 8.0e9; i++) launch_air_march(s_main, 1, (int)Nx - 1); // clocks up
      HIPCHK(hipStreamSynchronize(s_main));
      tune_ms[1] = timed([&] { launch_flips(s_main); launch_air_march(s_main, 1, (int)Nx - 1); });
      tune_ms[2] = 0.5f * timed([&] {
         launch_tb2(s_main, U0, U1, bufC, bufD);
         u0_src = U0; u1 = U1; u0 = bufC; launch_shell(s_main);
         u0_src = U1; u1 = bufC; u0 = bufD; launch_shell(s_main);
      });
      u0_src = nullptr; u0 = U0; u1 = U1;
      if (!(tune_ms[2] < pair_margin * tune_ms[1])) { // not worth it: drop the pair path and its two grids
         tb2 = false;
         for (Real *g : {bufC, bufD}) { own_list.erase(std::remove(own_list.begin(), own_list.end(), g), own_list.end()); hipFree(g); }
         bufC = bufD = nullptr;
      } else {
         HIPCHK(hipMemsetAsync(bufC, 0, npad * sizeof(Real), s_main));
         HIPCHK(hipMemsetAsync(bufD, 0, npad * sizeof(Real), s_main));
      }
      HIPCHK(hipDeviceSynchronize());
      hipEventDestroy(e0); hipEventDestroy(e1);
      return PF_OK;
   }
   // ---- where do the pair's four grids live?  The pair kernel streams four grids at once, and its speed depends on how
   // their PHYSICAL pages fall onto the memory channels relative to each other: engines of one process, alive side by side
   // and timed in turn, keep their own speed (1024^3: 3.07 / 3.07 / 3.79 ms per launch, round after round;
   // tools/placement_probe.py) -- a property of the allocations, not of the clock state, and nothing a virtual address
   // shows; nor do pairwise copy times between the grids, or any per-grid property: it is the combination that counts (a
   // pool of 8 grids at 1024^3: 2.94 ... 4.25 ms per launch over 64 assignments, a third of them within 2 % of the best).
   // So the engine samples: besides its own grids it allocates up to four more, times the pair kernel (both directions of
   // the four-grid cycle) on two dozen random assignments of pool members to the roles u^{n-1}, u^n, u^{n+1}, u^{n+2},
   // keeps the fastest and frees the rest.  A few hundred ms, once.  With caller-owned state grids (pf_opts.ext_u0 / ext_u1)
   // only the two spares are placed, and only the forward direction is timed (the caller's grids are not written).
   // the search itself: `evals` assignments of pool members to the roles (the first as given in `first`, the others drawn with
   // a fixed-seed generator), timed with the pair kernel; fixed_ab: the state grids stay where they are (u0 / u1), only the
   // two spares are drawn; both: the reverse direction of the four-grid cycle is timed as well (writes every member)
   int search_placement(const std::vector<Real *> &pool, bool fixed_ab, bool both, int evals, const int first[4], int chosen[4]) {
      const bool verbose = getenv("PFFDTD_VERBOSE") && atoi(getenv("PFFDTD_VERBOSE")) > 0;
      hipEvent_t e0, e1;
      HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
      const bool sampled = tb_sample && tb_nsample > 0;
      const float scale = sampled ? (float)(1.0 / tb_sample_frac) : 1.f; // sampled times are reported as whole-launch equivalents
      auto time_fwd = [&](Real *A, Real *B, Real *C, Real *D) -> float {
         hipEventRecord(e0, s_main);
         launch_probe(s_main, A, B, C, D, true);
         launch_probe(s_main, A, B, C, D, true);
         hipEventRecord(e1, s_main);
         hipEventSynchronize(e1);
         float ms = 0;
         hipEventElapsedTime(&ms, e0, e1);
         return ms / 2 * scale;
      };
      auto grid = [&](int i, Real *fallback) { return i >= 0 ? pool[i] : fallback; };
      for (int i = 0; i < 4; i++) launch_probe(s_main, grid(first[0], u0), grid(first[1], u1), pool[first[2]], pool[first[3]]); // clocks up
      struct Cand { int r[4]; float ms; };
      std::vector<Cand> cands;
      auto eval = [&](const int r[4]) {
         Real *A = grid(r[0], u0), *B = grid(r[1], u1), *C = pool[r[2]], *D = pool[r[3]];
         float ms = time_fwd(A, B, C, D);
         if (both) ms = 0.5f * (ms + time_fwd(C, D, A, B));
         cands.push_back({{r[0], r[1], r[2], r[3]}, ms});
      };
      uint32_t rng = 0x9e3779b9u;
      auto next = [&](uint32_t m) { rng = rng * 1664525u + 1013904223u; return (rng >> 8) % m; };
      eval(first);
      const int n = (int)pool.size(), k = fixed_ab ? 2 : 4;
      auto best_of = [&]() { size_t b = 0; for (size_t i = 1; i < cands.size(); i++) if (cands[i].ms < cands[b].ms) b = i; return b; };
      // phase 1: random assignments (a third of the budget); phase 2: from the best one, replace one role's grid at a time by
      // every other pool member (a swap when that member holds another role), keep what is faster, until a full sweep
      // brings nothing or the budget is spent.  Random draws alone reach the fastest level (one tuple in ~16) in three
      // pools of four; the descent gets there from the common second-best levels.
      const int nrand = std::max(evals / 3, 2);
      for (int t = 1; t < nrand && n >= k && (n > k || !fixed_ab); t++) {
         int idx[4], r[4] = {-1, -1, -1, -1};
         for (int i = 0; i < k; i++) { // k distinct pool members, in order
            bool dup;
            do { idx[i] = (int)next((uint32_t)n); dup = false; for (int j = 0; j < i; j++) dup |= idx[j] == idx[i]; } while (dup);
            r[4 - k + i] = idx[i];
         }
         eval(r);
      }
      for (bool improved = true; improved && (int)cands.size() < evals && n > k;) {
         improved = false;
         for (int role = 4 - k; role < 4 && (int)cands.size() < evals; role++) {
            for (int m = 0; m < n && (int)cands.size() < evals; m++) {
               const Cand cur = cands[best_of()];
               if (cur.r[role] == m) continue;
               int r[4] = {cur.r[0], cur.r[1], cur.r[2], cur.r[3]};
               for (int q = 4 - k; q < 4; q++) if (r[q] == m) r[q] = cur.r[role]; // m holds another role: swap
               r[role] = m;
               bool seen = false;
               for (auto &c : cands) seen |= c.r[0] == r[0] && c.r[1] == r[1] && c.r[2] == r[2] && c.r[3] == r[3];
               if (seen) continue;
               eval(r);
               if (cands.back().ms < 0.995f * cur.ms) improved = true;
            }
         }
      }
      size_t best = best_of();
      if (sampled) {
         // The sample ranks the candidates but over-states a launch by 5-10 % (fewer workgroups per launch, less halo sharing
         // in L2): the eight best are timed once more on ALL tiles and the fastest of those is kept; its whole-launch time
         // replaces the sample's (the caller compares it with the known fast level).
         std::vector<size_t> order(cands.size());
         std::iota(order.begin(), order.end(), 0);
         std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cands[a].ms < cands[b].ms; });
         float best_full = 0;
         std::vector<size_t> check(order.begin(), order.begin() + std::min<size_t>(8, order.size()));
         if (std::find(check.begin(), check.end(), (size_t)0) == check.end()) check.push_back(0); // (the as-allocated assignment, for the statistics)
         for (size_t k = 0; k < check.size(); k++) {
            Cand &c = cands[check[k]];
            Real *A = grid(c.r[0], u0), *B = grid(c.r[1], u1), *C = pool[c.r[2]], *D = pool[c.r[3]];
            auto full = [&](Real *a, Real *b, Real *cc, Real *d) {
               hipEventRecord(e0, s_main);
               launch_probe(s_main, a, b, cc, d); launch_probe(s_main, a, b, cc, d);
               hipEventRecord(e1, s_main); hipEventSynchronize(e1);
               float ms = 0; hipEventElapsedTime(&ms, e0, e1);
               return ms / 2;
            };
            float ms = full(A, B, C, D);
            if (both) ms = 0.5f * (ms + full(C, D, A, B));
            if (verbose) fprintf(stderr, "pffdtd_hip:   candidate %zu on all tiles: %.3f ms per launch (sample said %.3f)\n", check[k], ms, c.ms);
            c.ms = ms;
            if (k == 0 || ms < best_full) { best_full = ms; best = check[k]; }
         }
      }
      place_ms.clear();
      for (auto &c : cands) place_ms.push_back(c.ms);
      if (verbose) {
         fprintf(stderr, "pffdtd_hip: grid placement, %d candidates of a pool of %d%s:", (int)cands.size(), n, sampled ? " (timed on a sample of the tiles, scaled to a whole launch)" : "");
         for (size_t i = 0; i < cands.size(); i++) fprintf(stderr, " %.3f%s", cands[i].ms, i == best ? "*" : "");
         fprintf(stderr, " ms per launch\n");
      }
      for (int i = 0; i < 4; i++) chosen[i] = cands[best].r[i];
      hipEventDestroy(e0); hipEventDestroy(e1);
      return hipGetLastError() == hipSuccess ? PF_OK : set_err(PF_ERR_HIP, "placement search: kernel launch failed");
   }
   int place_evals() const {
      int evals = 48;
      return evals;
   }
   // the single-step paths stream two grids (u^n read, u^{n-1} read and overwritten): the same question with a smaller answer
   // (Musikverein, 13-point, 1.3e9 cells: 4.36-4.61 ms per step over the pairs of a pool of six; small grids: 1-3 %);
   // every unordered pair of the pool is timed on a whole step (both role assignments, the grids swap roles every step)
   int search_pair(const std::vector<Real *> &pool, int &bi, int &bj) {
      const bool verbose = getenv("PFFDTD_VERBOSE") && atoi(getenv("PFFDTD_VERBOSE")) > 0;
      hipEvent_t e0, e1;
      HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
      const int tsave = op.timing;
      op.timing = 0;
      Real *const s0 = u0, *const s1 = u1;
      auto one_step = [&](Real *a, Real *b) { // u0 = a, u1 = b; all-zero state: every kernel writes zeros
         u0 = a; u1 = b;
         launch_pre(s_main);
         launch_air(s_main, 1, (int)Nx - 1);
         launch_abc(s_main, {0, Nba});
         launch_rigid(s_main, {0, Nb});
         launch_fd(s_main, {0, Nbl});
      };
      auto time_pair = [&](Real *a, Real *b) -> float {
         hipEventRecord(e0, s_main);
         one_step(a, b); one_step(b, a); one_step(a, b); one_step(b, a);
         hipEventRecord(e1, s_main);
         hipEventSynchronize(e1);
         float ms = 0;
         hipEventElapsedTime(&ms, e0, e1);
         return ms / 4;
      };
      for (int i = 0; i < 6; i++) one_step(pool[0], pool[1]); // clocks up
      const int n = (int)pool.size();
      bi = 0; bj = 1;
      float best = 0, worst = 0, first = 0;
      place_ms.clear();
      for (int i = 0; i < n; i++)
         for (int j = i + 1; j < n; j++) {
            const float ms = time_pair(pool[i], pool[j]);
            place_ms.push_back(ms);
            if (i == 0 && j == 1) first = best = worst = ms;
            if (ms < best) { best = ms; bi = i; bj = j; }
            worst = std::max(worst, ms);
         }
      op.timing = tsave;
      u0 = s0; u1 = s1;
      if (verbose) fprintf(stderr, "pffdtd_hip: grid placement (single steps), %d pairs of a pool of %d: as allocated %.4f, chosen %.4f, slowest %.4f ms per step\n",
                           (int)place_ms.size(), n, first, best, worst);
      for (int i = 0; i < n; i++) HIPCHK(hipMemsetAsync(pool[i], 0, npad * sizeof(Real), s_main)); // (zeros from zeros; be explicit)
      HIPCHK(hipStreamSynchronize(s_main));
      hipEventDestroy(e0); hipEventDestroy(e1);
      return hipGetLastError() == hipSuccess ? PF_OK : set_err(PF_ERR_HIP, "placement search: kernel launch failed");
   }
   // Candidate grids beyond the engine's own: at most `want`, never more than fit the free memory (less 3 % of the device), and
   // never so many that the engine's grids plus the pool exceed 85 % of the device (1536^3 fp64: four 29 GB grids + four
   // candidates = 232 GB = 81 %; round 3 capped at 60 %, which left that engine one candidate and cost it 4 %: 217.7 vs 226 Gvox/s).
   // The candidates live for the search only.
   int pool_extra(int want, int own) const {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return want; }
      const double gb = (double)npad * sizeof(Real);
      const int cap = (int)std::floor((0.85 * (double)total_b - own * gb) / gb);
      const int fit = (int)std::floor(((double)free_b - 0.03 * (double)total_b) / gb);
      return std::max(0, std::min(want, std::min(cap, fit)));
   }
   bool place_single_ok() const { return !(op.debug & 0x8000) && !op.energy && vbase == 0 && npad * (int64_t)sizeof(Real) >= ((int64_t)64 << 20); }
   int sample_placement_single() {
      if (!own_grids || !place_single_ok()) return PF_OK;
      int extra = 4;
      extra = pool_extra(extra, 2);
      if (extra == 0) return PF_OK;
      std::vector<Real *> pool = {u0, u1};
      for (int i = 0; i < extra; i++) {
         Real *p = try_dzalloc<Real>(npad);
         if (!p) break;
         pool.push_back(p);
      }
      int bi, bj;
      int rc = search_pair(pool, bi, bj);
      if (rc) return rc;
      u0 = pool[bi]; u1 = pool[bj];
      for (int i = 0; i < (int)pool.size(); i++) {
         own_list.erase(std::remove(own_list.begin(), own_list.end(), pool[i]), own_list.end());
         if (i != bi && i != bj) hipFree(pool[i]);
      }
      own_list.push_back(u0); own_list.push_back(u1);
      return PF_OK;
   }
   int sample_placement() {
      if (!tb2 || tb2_slab || !bufC || !bufD || (op.debug & 0x8000) || vbase == 41) return PF_OK;
      int extra = tb3 ? 3 : 4;
      extra = pool_extra(extra, tb3 ? 5 : 4);
      if (extra == 0 && !own_grids) return PF_OK;
      std::vector<Real *> pool;
      if (own_grids) { pool.push_back(u0); pool.push_back(u1); }
      if (tb3) { // (roles 2, 3 = the grids k_tb3 writes; the fifth grid is one more candidate)
         if (!bufE) return PF_OK;
         pool.push_back(bufD); pool.push_back(bufE); pool.push_back(bufC);
      } else { pool.push_back(bufC); pool.push_back(bufD); }
      for (int i = 0; i < extra; i++) {
         Real *p = try_dzalloc<Real>(npad);
         if (!p) break; // no room for another candidate
         pool.push_back(p);
      }
      const int first_own[4] = {0, 1, 2, 3}, first_ext[4] = {-1, -1, 0, 1};
      int w[4];
      int rc = search_placement(pool, !own_grids, own_grids, place_evals(), own_grids ? first_own : first_ext, w);
      if (rc) return rc;
      // Some pools hold no fast assignment at all (seen once in ~15 boxes: best 3.37 ms of 47 candidates where 2.95 is the
      // rule).  For the 7-point kernel the fast level is known -- the compulsory bytes of a pair, 4 grids x cells, at
      // 5.5 TB/s -- so a search that ends well above it gets four more grids to choose from, twice at most.
      const float as_allocated = place_ms.empty() ? 0.f : place_ms[0];
      for (int round = 0; round < 2 && !fcc && !tb3 && own_grids; round++) {
         const float best = *std::min_element(place_ms.begin(), place_ms.end());
         const float target = (float)((double)tb_clean_cells * 4.0 * sizeof(Real) / 5.5e12 * 1e3);
         if (best <= 1.05f * target) break;
         size_t grown = 0;
         for (int i = 0, more = pool_extra(4, (int)pool.size()); i < more; i++) {
            Real *p = try_dzalloc<Real>(npad);
            if (!p) break;
            pool.push_back(p); grown++;
         }
         if (!grown) break;
         const int cur[4] = {w[0], w[1], w[2], w[3]};
         if ((rc = search_placement(pool, false, true, place_evals(), cur, w))) return rc;
         place_ms.insert(place_ms.begin(), as_allocated); // (the statistics keep the very first candidate in front)
      }
      std::vector<Real *> keep;
      if (own_grids) { u0 = pool[w[0]]; u1 = pool[w[1]]; keep.push_back(u0); keep.push_back(u1); }
      if (tb3) { // the four streams of k_tb3 are u0, u1 -> bufD, bufE; bufC (u^{n+1} of the shell and of a few tiles) is any other member
         bufD = pool[w[2]]; bufE = pool[w[3]];
         bufC = nullptr;
         for (Real *g : pool)
            if (g != u0 && g != u1 && g != bufD && g != bufE) { bufC = g; break; }
         if (!bufC) return set_err(PF_ERR_STATE, "placement search: no fifth grid left"); // (the pool holds the engine's five)
         keep.push_back(bufC); keep.push_back(bufD); keep.push_back(bufE);
      } else {
      bufC = pool[w[2]]; bufD = pool[w[3]];
      keep.push_back(bufC); keep.push_back(bufD);
      }
      for (Real *g : pool) {
         own_list.erase(std::remove(own_list.begin(), own_list.end(), g), own_list.end());
         if (std::find(keep.begin(), keep.end(), g) == keep.end()) hipFree(g);
      }
      for (Real *g : keep) {
         own_list.push_back(g);
         HIPCHK(hipMemsetAsync(g, 0, npad * sizeof(Real), s_main)); // (the pair kernel wrote zeros computed from zeros; be explicit)
      }
      HIPCHK(hipStreamSynchronize(s_main));
      return PF_OK;
   }
   // Slab engines (caller-owned grids): the caller offers a pool of n >= 4 zero-filled grids before the first step; the
   // engine adopts the fastest assignment of four of them -- idx[0], idx[1]: the state grids (they replace ext_u0 / ext_u1),
   // idx[2], idx[3]: the spares of pf_engine_set_spares, or idx = 0, 1, -1, -1 when this engine steps singly.
   int place_grids(void *const *grids, int n, int32_t *idx) override { return place_grids_impl(grids, n, idx, false); }
   int place_grids5(void *const *grids, int n, int32_t *idx) override { return place_grids_impl(grids, n, idx, true); }
   // five: idx has five entries and the engine may step in TRIPLES across three split-phase steps (idx[2], idx[3] = the grids k_tb3
   // writes, idx[4] = the u^{n+1} grid; idx[4] = -1: pairs or single steps as pf_engine_place_grids reports them)
   int place_grids_impl(void *const *grids, int n, int32_t *idx, bool five) {
      if (in_step || pair_phase || steps_done > 0) return set_err(PF_ERR_STATE, "pf_engine_place_grids after the first step");
      if (state_touched) return set_err(PF_ERR_STATE, "pf_engine_place_grids after pf_engine_set_grid: the placement search runs step kernels on the offered grids and zeroes them");
      if (own_grids) return set_err(PF_ERR_STATE, "pf_engine_place_grids: this engine allocated its own grids");
      if (!grids || !idx || n < 2) return set_err(PF_ERR_ARG, "pf_engine_place_grids: need a pool of at least two grids");
      for (int i = 0; i < n; i++) {
         if (!grids[i]) return set_err(PF_ERR_ARG, "pf_engine_place_grids: null grid");
         for (int j = 0; j < i; j++) if (grids[i] == grids[j]) return set_err(PF_ERR_ARG, "pf_engine_place_grids: grid offered twice");
      }
      HIPCHK(hipSetDevice(op.device));
      u0 = (Real *)grids[0]; u1 = (Real *)grids[1];
      idx[0] = 0; idx[1] = 1; idx[2] = idx[3] = -1;
      if (five) idx[4] = -1;
      if (five && n >= 5 && tb3_geom && tb2_geom && !(op.slab_first && op.slab_last)) {
         // triples: the wall regions must fit the triples' box (init_walls re-orders the lossy arrays: before the first step only)
         tb2_slab = true;
         if (!(op.debug & 0x10000000)) { int rcw = init_walls(true); if (rcw) return rcw; }
         if (wl_on) {
            std::vector<Real *> pool;
            for (int i = 0; i < n; i++) pool.push_back((Real *)grids[i]);
            int w[4] = {0, 1, 2, 3};
            if (n > 5 && !(op.debug & 0x8000)) {
               const int first[4] = {0, 1, 2, 3};
               tb2_probe = true;
               int rc = search_placement(pool, false, true, place_evals(), first, w);
               tb2_probe = false;
               if (rc) return rc;
               for (int i = 0; i < n; i++) HIPCHK(hipMemsetAsync(pool[i], 0, npad * sizeof(Real), s_main));
               HIPCHK(hipStreamSynchronize(s_main));
            }
            int c = -1;
            for (int i = 0; i < n && c < 0; i++) if (i != w[0] && i != w[1] && i != w[2] && i != w[3]) c = i;
            u0 = pool[w[0]]; u1 = pool[w[1]]; bufD = pool[w[2]]; bufE = pool[w[3]]; bufC = pool[c];
            for (int i = 0; i < 4; i++) idx[i] = w[i];
            idx[4] = c;
            tb3_slab = true;
            tb3_remember_home();
            return PF_OK;
         }
         tb2_slab = false; // (the regions do not fit the triples' box: pairs)
      }
      if (steps_done == 0) { int rcg = pairs_geometry(); if (rcg) return rcg; }
      if (n < 4 || !tb2_geom || (op.slab_first && op.slab_last)) { // keeps stepping singly: on the fastest pair of the pool
         if (n > 2 && place_single_ok()) {
            std::vector<Real *> pool;
            for (int i = 0; i < n; i++) pool.push_back((Real *)grids[i]);
            int bi, bj;
            int rc = search_pair(pool, bi, bj);
            if (rc) return rc;
            u0 = pool[bi]; u1 = pool[bj];
            idx[0] = bi; idx[1] = bj;
         }
         return PF_OK;
      }
      std::vector<Real *> pool;
      for (int i = 0; i < n; i++) pool.push_back((Real *)grids[i]);
      int w[4] = {0, 1, 2, 3};
      if (n > 4 && !(op.debug & 0x8000)) {
         const int first[4] = {0, 1, 2, 3};
         tb2_probe = true;
         int rc = search_placement(pool, false, true, place_evals(), first, w);
         tb2_probe = false;
         if (rc) return rc;
         for (int i = 0; i < n; i++) HIPCHK(hipMemsetAsync(pool[i], 0, npad * sizeof(Real), s_main));
         HIPCHK(hipStreamSynchronize(s_main));
      }
      u0 = pool[w[0]]; u1 = pool[w[1]]; bufC = pool[w[2]]; bufD = pool[w[3]];
      for (int i = 0; i < 4; i++) idx[i] = w[i];
      tb2_slab = true;
      if (!wl_on && !(op.debug & 0x10000000)) { int rcw = init_walls(true); if (rcw) return rcw; }
      return PF_OK;
   }
   int autotune() {
      if (fcc) return autotune_fcc();
      if (vbase != 0 || fcc || op.energy || (op.debug & 0x8000) || !use_dpp || !(lean || vg)) return PF_OK;
      if (Nx * Ny * Nz < ((int64_t)1 << 22)) return PF_OK; // tiny grids: launch-bound either way
      Real *scr = bufC;
      bool own = false;
      int rc;
      if (!scr) { scr = try_dzalloc<Real>(npad); if (!scr) return PF_OK; own = true; } // no room to measure: the static rules stand
      hipEvent_t e0, e1;
      HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
      HIPCHK(hipDeviceSynchronize());
      auto timed = [&](auto &&fn) -> float {
         fn();
         hipEventRecord(e0, s_main);
         for (int i = 0; i < 3; i++) fn();
         hipEventRecord(e1, s_main);
         hipEventSynchronize(e1);
         float ms = 0;
         hipEventElapsedTime(&ms, e0, e1);
         return ms / 3;
      };
      const bool lean0 = lean, vg0 = vg;
      Real *U0 = u0, *U1 = u1;
      lean = true; vg = false; u0_src = U0; u0 = scr;
      // the device has been idle while the host built the lists: ramp its clocks first (~20 ms of work), or the first
      // candidate is measured -- and every launch here profiled -- at idle clocks (seen: +56 % per launch)
      for (int i = 0; i < 8 && (double)i * (double)(Nx * Ny * Nz) < 8.0e9; i++) launch_air_lean(s_main, 1, (int)Nx - 1);
      HIPCHK(hipStreamSynchronize(s_main));
      tune_ms[0] = timed([&] { launch_air_lean(s_main, 1, (int)Nx - 1); });
      u0 = U0; u0_src = nullptr;
      lean = false; vg = true; v1_dst = scr;
      { // the barrier-free kernel, with 64 / 32 / 16 lanes per row segment where those pad the rows differently
         constexpr int V = pf::VecOf<Real>::V;
         int best_lw = 0;
         int64_t seen[3] = {0, 0, 0};
         int k = 0;
         for (int lw : {64, 32, 16}) {
            const int64_t w = cdiv(P, (int64_t)lw * V) * lw * V;
            if (k > 0 && w == seen[k - 1]) continue; // same padded width as the wider segment: the wider one wins anyway
            seen[k++] = w;
            lw_force = lw;
            const float t = timed([&] { launch_air_march(s_main, 1, (int)Nx - 1); });
            if (best_lw == 0 || t < 0.98f * tune_ms[1]) { tune_ms[1] = t; best_lw = lw; }
         }
         lw_force = best_lw;
      }
      v1_dst = nullptr;
      lean = lean0; vg = vg0;
      if (hipGetLastError() != hipSuccess) { lean = lean0; vg = vg0; }
      else if (tune_ms[1] < 0.97f * tune_ms[0]) { lean = false; vg = true; }
      else if (tune_ms[0] < 0.97f * tune_ms[1]) { lean = true; vg = false; }
      if (tb3) {
         // three steps per pass, the boundary pass included: the single steps get theirs added (fields and branch state are all zeros
         // at creation and stay so)
         u0_src = U0; u0 = scr;
         const float tb = timed([&] { launch_rigid(s_main, {0, Nb}); });
         u0_src = nullptr; u0 = U0;
         tune_ms[0] += tb; tune_ms[1] += tb;
         tune_ms[2] = (1.f / 3.f) * timed([&] {
            bnd_sel = wl_rest;
            u0_src = U0; u1 = U1; u0 = bufC; launch_dirty_tiles(s_main); launch_rigid(s_main, {0, wl_nrest});
            launch_walls(s_main, s_main, U0, U1, bufC, bufD, ub[0], ub[1], ub[2]);
            launch_tb3(s_main, U0, U1, bufC, bufD, bufE);
            u0_src = U1; u1 = bufC; u0 = bufD; launch_dirty_tiles(s_main); launch_rigid(s_main, {0, wl_nrest});
            bnd_sel = nullptr;
            u0_src = bufC; u1 = bufD; u0 = bufE; launch_shell(s_main); launch_rigid(s_main, {0, Nb});
            u0_src = nullptr; u0 = U0; u1 = U1;
         });
      } else if (tb2 && wl_on) {
         // wall regions: the pair then includes the boundary pass, so the single steps get theirs added (fields and branch state
         // are all zeros at creation and stay so)
         u0_src = U0; u0 = scr;
         const float tb = timed([&] { launch_rigid(s_main, {0, Nb}); });
         u0_src = nullptr; u0 = U0;
         tune_ms[0] += tb; tune_ms[1] += tb;
         tune_ms[2] = 0.5f * timed([&] {
            launch_tb2(s_main, U0, U1, bufC, bufD);
            bnd_sel = wl_rest;
            u0_src = U0; u1 = U1; u0 = bufC; launch_dirty_tiles(s_main); launch_rigid(s_main, {0, wl_nrest});
            launch_walls(s_main, s_main, U0, U1, bufC, bufD, ub[0], ub[1], ub[2]);
            u0_src = U1; u1 = bufC; u0 = bufD; launch_dirty_tiles(s_main); launch_rigid(s_main, {0, wl_nrest});
            bnd_sel = nullptr;
            u0_src = nullptr; u0 = U0; u1 = U1;
         });
      } else if (tb2) {
         tune_ms[2] = 0.5f * timed([&] {
            launch_tb2(s_main, U0, U1, bufC, bufD);
            u0_src = U0; u1 = U1; u0 = bufC; launch_shell(s_main);
            u0_src = U1; u1 = bufC; u0 = bufD; launch_shell(s_main);
            u0_src = nullptr; u0 = U0; u1 = U1;
         });
      }
      if (tb2) {
         if (!(tune_ms[2] < pair_margin * std::min(tune_ms[0], tune_ms[1]))) { // not worth it: drop the blocked path and its extra grids
            if (scr == bufC) scr = nullptr;
            drop_blocking();
         } else {
            HIPCHK(hipMemsetAsync(bufC, 0, npad * sizeof(Real), s_main));
            HIPCHK(hipMemsetAsync(bufD, 0, npad * sizeof(Real), s_main));
            if (bufE) HIPCHK(hipMemsetAsync(bufE, 0, npad * sizeof(Real), s_main));
         }
      }
      HIPCHK(hipDeviceSynchronize());
      hipEventDestroy(e0); hipEventDestroy(e1);
      if (own && scr) hipFree(scr);
      return PF_OK;
   }
   pf::Tb2Params tile_params() const {
      pf::Tb2Params tp{};
      tp.plane = plane; tp.Nx = (int)Nx; tp.Ny = (int)Ny; tp.Nz = (int)Nz; tp.P = (int)P;
      tp.x_begin = tbx0; tp.x_end = tbx1; tp.y_begin = tby0; tp.y_end = tby1; tp.z_begin = tbz0; tp.z_end = tbz1;
      tp.chunk = tb_chunk; tp.nxc = tb_nxc; tp.nyt = tb_nyt; tp.nzt = tb_nzt;
      return tp;
   }
   // two steps of the clean tiles
   // three steps of the clean tiles: A = u^{n-1}, B = u^n -> D = u^{n+2}, E = u^{n+3}; C: where flagged tiles leave their u^{n+1} (null: nowhere)
   void launch_tb3(hipStream_t s, const Real *A, const Real *B, Real *C, Real *D, Real *E, bool sample = false) {
      if (tb_xr.empty() || tb_nclean <= 0) return;
      pf::Tb2Params tp = tile_params();
      tp.A = A; tp.B = B; tp.C = C; tp.D = D; tp.E = E;
      if (!(op.slab_first && op.slab_last)) tp.band |= 2; // a slab: the planes beside the box read the u^{n+1} of its first and last plane
      sample = sample && tb_sample && tb_nsample > 0;
      tp.tiles = sample ? tb_sample : tb_clean;
      const dim3 g((uint32_t)(sample ? tb_nsample : tb_nclean)), b(64 * tb3_wt);
      if (sg) {
         if (tb2_probe) hipLaunchKernelGGL((pf::k_tb3<Real, tb3_r, tb3_wt, true, true>), g, b, 0, s, tp, a1, a2);
         else hipLaunchKernelGGL((pf::k_tb3<Real, tb3_r, tb3_wt, true, false>), g, b, 0, s, tp, a1, a2);
      } else {
         if (tb2_probe) hipLaunchKernelGGL((pf::k_tb3<Real, tb3_r, tb3_wt, false, true>), g, b, 0, s, tp, a1, a2);
         else hipLaunchKernelGGL((pf::k_tb3<Real, tb3_r, tb3_wt, false, false>), g, b, 0, s, tp, a1, a2);
      }
   }
   // the blocked kernel as the creation-time measurements see it: its four streams (k_tb3: two grids read, two written)
   void launch_probe(hipStream_t s, const Real *A, const Real *B, Real *C, Real *D, bool sample = false) {
      if (triples() || tb3_geom) launch_tb3(s, A, B, nullptr, C, D, sample);
      else launch_tb2(s, A, B, C, D, sample);
   }
   void launch_tb2(hipStream_t s, const Real *A, const Real *B, Real *C, Real *D, bool sample = false) {
      if (triples()) { // a pair on the triples' tiles: k_tb3's two-step form (the last two steps of a run, Engine::run)
         if (tb_xr.empty() || tb_nclean <= 0) return;
         pf::Tb2Params tp = tile_params();
         tp.A = A; tp.B = B; tp.C = C; tp.D = D; tp.E = nullptr;
         tp.tiles = tb_clean;
         const dim3 g((uint32_t)tb_nclean), b(64 * tb3_wt);
         if (sg) hipLaunchKernelGGL((pf::k_tb3<Real, tb3_r, tb3_wt, true, false, 2>), g, b, 0, s, tp, a1, a2);
         else hipLaunchKernelGGL((pf::k_tb3<Real, tb3_r, tb3_wt, false, false, 2>), g, b, 0, s, tp, a1, a2);
         return;
      }
      if (tb_xr.empty() || tb_nclean <= 0) return;
      pf::Tb2Params tp = tile_params();
      tp.A = A; tp.B = B; tp.C = C; tp.D = D;
      tp.tiles = (tb_ndirty > 0 || tb_order_band) ? tb_clean : nullptr; // all clean: the dense order (identical to the list's)
      sample = sample && tb_sample && tb_nsample > 0;
      if (sample) tp.tiles = tb_sample;
      const dim3 g((uint32_t)(sample ? tb_nsample : tb_nclean)), b(256);
      if (fcc) {
         if (tb_lw == 64) { // 12-row tiles; k_tb2_fcc_w: half the vector arithmetic of k_tb2_fcc_x (debug 0x40000: that one, CPU-exact file order only)
            if ((op.debug & 0x40000) && !sg && !swz) hipLaunchKernelGGL((pf::k_tb2_fcc_x<Real, 2, 8>), g, dim3(512), 0, s, tp, a1, a2);
            else if (sg) { if (swz) hipLaunchKernelGGL((pf::k_tb2_fcc_w<Real, 2, 8, true, true>), g, dim3(512), 0, s, tp, a1, a2);
                           else hipLaunchKernelGGL((pf::k_tb2_fcc_w<Real, 2, 8, true, false>), g, dim3(512), 0, s, tp, a1, a2); }
            else { if (swz) hipLaunchKernelGGL((pf::k_tb2_fcc_w<Real, 2, 8, false, true>), g, dim3(512), 0, s, tp, a1, a2);
                   else hipLaunchKernelGGL((pf::k_tb2_fcc_w<Real, 2, 8, false, false>), g, dim3(512), 0, s, tp, a1, a2); }
         } else pf::launch_tb2_fcc<Real>(s, tp, a1, a2, tb_lw, (uint32_t)(sample ? tb_nsample : tb_nclean), sg, swz);
         return;
      }
      if (swz) { // axes exchanged in storage (64-lane segments only, init_tb2)
         if (sg) hipLaunchKernelGGL((pf::k_tb2_reg<Real, 3, 4, false, 64, false, true, true>), g, b, 0, s, tp, a1, a2);
         else hipLaunchKernelGGL((pf::k_tb2_reg<Real, 3, 4, false, 64, false, false, true>), g, b, 0, s, tp, a1, a2);
         return;
      }
      if (sg) { // the reference GPU engine's arithmetic (towards-zero pairwise sums, two FMAs)
         if (tb_lw == 32) hipLaunchKernelGGL((pf::k_tb2_reg<Real, 3, 4, false, 32, false, true>), g, b, 0, s, tp, a1, a2);
         else if (tb_lw == 16) hipLaunchKernelGGL((pf::k_tb2_reg<Real, 3, 4, false, 16, false, true>), g, b, 0, s, tp, a1, a2);
         else if (tb2_probe) hipLaunchKernelGGL((pf::k_tb2_reg<Real, 3, 4, false, 64, true, true>), g, b, 0, s, tp, a1, a2);
         else hipLaunchKernelGGL((pf::k_tb2_reg<Real, 3, 4, false, 64, false, true>), g, b, 0, s, tp, a1, a2);
         return;
      }
      if (tb_lw == 32) hipLaunchKernelGGL((pf::k_tb2_reg<Real, 3, 4, false, 32>), g, b, 0, s, tp, a1, a2);
      else if (tb_lw == 16) hipLaunchKernelGGL((pf::k_tb2_reg<Real, 3, 4, false, 16>), g, b, 0, s, tp, a1, a2);
      else if (tb2_probe) hipLaunchKernelGGL((pf::k_tb2_reg<Real, 3, 4, false, 64, true>), g, b, 0, s, tp, a1, a2);
      else hipLaunchKernelGGL((pf::k_tb2_reg<Real, 3, 4, false, 64>), g, b, 0, s, tp, a1, a2);
   }
   // one out-of-place step of the dirty tiles: u1, (u0_src old) -> u0
   void launch_dirty_tiles(hipStream_t s) {
      if (tb_xr.empty() || tb_ndirty <= 0) return;
      pf::Tb2Params tp = tile_params();
      tp.A = u0_src ? u0_src : u0; tp.B = u1; tp.C = u0; tp.D = nullptr;
      tp.tiles = tb_dirty; tp.mask = mask;
      tp.xsub = tb_ndirty <= 256 ? std::min(16, std::max(tb_chunk / 4, 1)) : 1; // few tiles: shorter marches (4 planes), more workgroups
      const dim3 g((uint32_t)tb_ndirty * (uint32_t)tp.xsub), b(256);
      if (fcc) return; // (13-point: k_air_fcc over its own tiling of the box's planes, sh_tiles)
      if (triples() || tb3_geom) { // k_tb3's tiles: 20 rows
         static_assert(tb3_rows == 20, "k_tb1_tile<Real, 5, 4>: 20-row tiles");
         constexpr int HL = pf::VecOf<Real>::V >= 3 ? 1 : 2; // (the tiles' column ranges are k_tb3's)
         if (sg) hipLaunchKernelGGL((pf::k_tb1_tile<Real, 5, 4, 64, true, false, HL>), g, b, 0, s, tp, a1, a2);
         else hipLaunchKernelGGL((pf::k_tb1_tile<Real, 5, 4, 64, false, false, HL>), g, b, 0, s, tp, a1, a2);
         return;
      }
      if (swz) {
         if (sg) hipLaunchKernelGGL((pf::k_tb1_tile<Real, 3, 4, 64, true, true>), g, b, 0, s, tp, a1, a2);
         else hipLaunchKernelGGL((pf::k_tb1_tile<Real, 3, 4, 64, false, true>), g, b, 0, s, tp, a1, a2);
         return;
      }
      if (sg) {
         if (tb_lw == 32) hipLaunchKernelGGL((pf::k_tb1_tile<Real, 3, 4, 32, true>), g, b, 0, s, tp, a1, a2);
         else if (tb_lw == 16) hipLaunchKernelGGL((pf::k_tb1_tile<Real, 3, 4, 16, true>), g, b, 0, s, tp, a1, a2);
         else hipLaunchKernelGGL((pf::k_tb1_tile<Real, 3, 4, 64, true>), g, b, 0, s, tp, a1, a2);
         return;
      }
      if (tb_lw == 32) hipLaunchKernelGGL((pf::k_tb1_tile<Real, 3, 4, 32>), g, b, 0, s, tp, a1, a2);
      else if (tb_lw == 16) hipLaunchKernelGGL((pf::k_tb1_tile<Real, 3, 4, 16>), g, b, 0, s, tp, a1, a2);
      else hipLaunchKernelGGL((pf::k_tb1_tile<Real, 3, 4, 64>), g, b, 0, s, tp, a1, a2);
   }
   // one out-of-place single step of everything outside the box: u1 -> (u0_src old) -> u0
   void launch_shell(hipStream_t s) { launch_shell(s, 1, (int)Nx - 1); }
   // 13-point: ghost flips of u1 in memory, then x slabs (whole planes), column strips, and the single-step tiles of the box
   void launch_shell_fcc(hipStream_t s, int xlo, int xhi, bool flips) { // planes [xlo, xhi); flips: also the ghost shell of u1
      if (flips) launch_flips(s);
      if (tb_xr.empty()) { launch_air_march(s, xlo, xhi); return; }
      if (tbx0 > xlo) launch_air_march(s, xlo, tbx0);
      if (tbx1 < xhi) launch_air_march(s, tbx1, xhi);
      constexpr int V = pf::VecOf<Real>::V;
      {
         pf::ZStripParams<Real> zp{};
         zp.u1 = u1; zp.u0s = u0_src ? u0_src : u0; zp.u0 = u0; zp.mask = mask;
         zp.plane = plane; zp.Nx = (int)Nx; zp.Ny = (int)Ny; zp.Nz = (int)Nz; zp.P = (int)P;
         zp.x_begin = tbx0; zp.x_end = tbx1; zp.zl = tbz0; zp.zr = tbz1; zp.first = op.slab_first; zp.last = op.slab_last;
         const int64_t nthreads = (int64_t)(zp.zl / V + (P - zp.zr) / V) * (Ny - 2);
         const int xchunk = 16;
         const dim3 gz((unsigned)cdiv(nthreads, 256), (unsigned)cdiv(tbx1 - tbx0, xchunk));
         if (nthreads > 0) {
            if (sg) { if (swz) hipLaunchKernelGGL((pf::k_zstrip_fcc<Real, true, true>), gz, dim3(256), 0, s, zp, a1, a2, l, xchunk, fold ? 1 : 0);
                      else hipLaunchKernelGGL((pf::k_zstrip_fcc<Real, true, false>), gz, dim3(256), 0, s, zp, a1, a2, l, xchunk, fold ? 1 : 0); }
            else { if (swz) hipLaunchKernelGGL((pf::k_zstrip_fcc<Real, false, true>), gz, dim3(256), 0, s, zp, a1, a2, l, xchunk, fold ? 1 : 0);
                   else hipLaunchKernelGGL((pf::k_zstrip_fcc<Real, false, false>), gz, dim3(256), 0, s, zp, a1, a2, l, xchunk, fold ? 1 : 0); }
         }
      }
      if (sh_ntiles > 0) {
         pf::AirParams ap;
         ap.Ny = Ny; ap.P = P; ap.plane = plane;
         ap.x_begin = tbx0; ap.x_end = tbx1; ap.chunk = tb_chunk; ap.nxc = tb_nxc; ap.nzt = sh_nzt; ap.nyt = sh_nyt;
         ap.swizzle = 0; ap.swz = swz ? 1 : 0;
         ap.Nx = (int)Nx; ap.Nz = (int)Nz; ap.first = op.slab_first; ap.last = op.slab_last; ap.fold = fold ? 1 : 0;
         if (sg) hipLaunchKernelGGL((pf::k_air_fcc<Real, 4, 4, 1, true, true, false, true, 64>), dim3((uint32_t)sh_ntiles), dim3(256), 0, s, u1, u0, mask, a1, a2,
                                    ap, l, u0_src, sh_tiles);
         else hipLaunchKernelGGL((pf::k_air_fcc<Real, 4, 4, 1, false, true, false, true, 64>), dim3((uint32_t)sh_ntiles), dim3(256), 0, s, u1, u0, mask, a1, a2,
                            ap, l, u0_src, sh_tiles);
      }
      launch_dirty_tiles(s);
   }
   // planes [xlo, xhi) without the strips beside the box (slab pairs with wall regions): whole planes outside the box's x range
   // and the box's own single-step tiles
   void launch_shell_planes(hipStream_t s, int xlo, int xhi, bool tiles = true) {
      int xa = xlo;
      for (auto &r : tb_xr) {
         if (r.first > xa) launch_air_lean(s, xa, r.first);
         xa = std::max(xa, r.second);
      }
      if (xhi > xa) launch_air_lean(s, xa, xhi);
      if (tiles) launch_dirty_tiles(s);
   }
   void launch_shell(hipStream_t s, int xlo, int xhi) { // planes [xlo, xhi) (the box lies inside)
      if (fcc) { launch_shell_fcc(s, xlo, xhi, !tb2_slab); return; } // (slab engines flip on the edge stream, after the exchange)
      int xa = xlo;
      if (tb_xr.size() == 1 && (vbase == 0 || vbase == 40 || vbase == 41) && tb_xr[0].first > xlo && xhi > tb_xr[0].second &&
          tb_xr[0].first - xlo <= 16 && xhi - tb_xr[0].second <= 16) {
         // the usual case: two thin x slabs, below and above the box -- one launch instead of two latency-bound ones
         lean_x2_begin = tb_xr[0].second; lean_x2_end = xhi;
         launch_air_lean(s, xlo, tb_xr[0].first);
         lean_x2_begin = lean_x2_end = 0;
         xa = xhi;
      }
      for (auto &r : tb_xr) { // x slabs: everything before / between / after the box's plane ranges, full planes
         if (r.first > xa) launch_air_lean(s, xa, r.first);
         xa = std::max(xa, r.second);
      }
      if (xhi > xa) launch_air_lean(s, xa, xhi);
      if (tb_xr.empty()) return;
      const int xb = tb_xr.front().first, xe = tb_xr.back().second;
      // beside the box: the two row strips in one lean launch (tile height of the default configuration: 16 rows) ...
      const int th = 8; // lean<2,4>: 8-row tiles (the strips are 5-7 rows thick in a box-shaped room)
      lean_nyt = (int)cdiv(tby0 - 1, th); lean_yt0 = (tby1 - 1) / th;
      launch_lean_cfg<2, 4>(s, xb, xe);
      lean_nyt = -1; lean_yt0 = 0;
      // ... and the two column strips
      {
         constexpr int V = pf::VecOf<Real>::V;
         pf::ZStripParams<Real> zp{};
         zp.u1 = u1; zp.u0s = u0_src ? u0_src : u0; zp.u0 = u0; zp.mask = mask;
         zp.plane = plane; zp.Nx = (int)Nx; zp.Ny = (int)Ny; zp.Nz = (int)Nz; zp.P = (int)P;
         zp.x_begin = xb; zp.x_end = xe; zp.zl = szl; zp.zr = szr; zp.first = op.slab_first; zp.last = op.slab_last;
         if (zs_map && bnd_sel) { // (inside step_pair) the strips' boundary nodes are updated right here
            zp.zvec = zs_map; zp.adjv = zs_adj; zp.lossy = zs_li; zp.u0b = ub[0]; zp.sl2 = sl2;
         }
         const int64_t nthreads = (int64_t)(zp.zl / V + (P - zp.zr) / V) * (Ny - 2);
         const int xchunk = 16;
         const dim3 gz((unsigned)cdiv(nthreads, 256), (unsigned)cdiv(xe - xb, xchunk));
         if (swz) { if (sg) hipLaunchKernelGGL((pf::k_air_zstrip<Real, true, true>), gz, dim3(256), 0, s, zp, a1, a2, l, xchunk);
                    else hipLaunchKernelGGL((pf::k_air_zstrip<Real, false, true>), gz, dim3(256), 0, s, zp, a1, a2, l, xchunk); }
         else if (sg) hipLaunchKernelGGL((pf::k_air_zstrip<Real, true>), gz, dim3(256), 0, s, zp, a1, a2, l, xchunk);
         else hipLaunchKernelGGL((pf::k_air_zstrip<Real, false>), gz, dim3(256), 0, s, zp, a1, a2, l, xchunk);
      }
      launch_dirty_tiles(s); // ... and the tiles of the box that hold geometry or a source
   }
   // steps n and n+1 in one go; the state moves from (u0, u1) to (bufC, bufD), which swap roles with them
   int step_pair(int64_t n) {
      if (n < 0 || n + 1 >= Nt) return set_err(PF_ERR_ARG, "step pair %ld outside [0,Nt=%ld)", (long)n, (long)Nt);
      if (wl_on) return step_pair_walls(n);
      hipStream_t s = s_main;
      Real *A = u0, *B = u1, *C = bufC, *D = bufD;
      std::pair<hipEvent_t, hipEvent_t> ev{}, eva{};
      auto get_ev = [&]() { std::pair<hipEvent_t, hipEvent_t> e{}; if (!ev_pool.empty()) { e = ev_pool.back(); ev_pool.pop_back(); } else { hipEventCreate(&e.first); hipEventCreate(&e.second); } return e; };
      if (op.timing) { ev = get_ev(); eva = get_ev(); hipEventRecord(ev.first, s); hipEventRecord(eva.first, s); } // step events: one per step
      std::pair<hipEvent_t, hipEvent_t> evt{};
      // The FIRST step of the shell reads u^{n-1}, u^n only and writes cells the pair kernel does not: it runs BESIDE the pair
      // kernel, on the edge stream (its launches are small and latency-bound -- strided strips, list gathers -- and fill the gaps
      // the bandwidth-bound pair kernel leaves); the second step needs the box's u^{n+1} and follows.  debug 0x4000000: one stream.
      const bool beside = !(op.debug & 0x4000000);
      hipStream_t sh = beside ? s_edge : s;
      if (beside) { HIPCHK(hipEventRecord(ev_pre, s)); HIPCHK(hipStreamWaitEvent(s_edge, ev_pre, 0)); }
      u0_src = A; u1 = B; u0 = C;
      const Range bnd = zs_map ? Range{0, zs_nrest} : Range{0, Nb};
      bnd_sel = zs_map ? zs_rest : nullptr;
      launch_shell(sh);
      launch_rigid(sh, bnd);
      launch_fd(sh, {0, Nbl});
      launch_io(sh, n, true, {0, Ns});
      // (with per-launch events on, the pair kernel waits for the shell: its recorded duration is the kernel's own, not the overlap's)
      if (op.timing && beside) { HIPCHK(hipEventRecord(ev_edge, s_edge)); HIPCHK(hipStreamWaitEvent(s, ev_edge, 0)); }
      if (op.timing) { evt = get_ev(); hipEventRecord(evt.first, s); }
      launch_tb2(s, A, B, C, D);
      if (op.timing) { hipEventRecord(evt.second, s); tb2_ev.push_back(evt); }
      if (beside && !op.timing) { HIPCHK(hipEventRecord(ev_edge, s_edge)); HIPCHK(hipStreamWaitEvent(s, ev_edge, 0)); }
      if (op.timing) { hipEventRecord(eva.second, s); air_ev.push_back(eva); eva = get_ev(); } // ("air" of the first step: the pair kernel and the shell beside it)
      { Real *t = ub[2]; ub[2] = ub[1]; ub[1] = ub[0]; ub[0] = t; }
      if (ring_fill == 0) ring_n0 = n;
      ring_fill++; steps_done++;
      u0_src = B; u1 = C; u0 = D;
      if (op.timing) { hipEventRecord(ev.second, s); step_ev.push_back(ev); ev = get_ev(); hipEventRecord(ev.first, s); hipEventRecord(eva.first, s); }
      launch_shell(s);
      if (op.timing) { hipEventRecord(eva.second, s); air_ev.push_back(eva); }
      launch_rigid(s, bnd);
      bnd_sel = nullptr;
      launch_fd(s, {0, Nbl});
      launch_io(s, n + 1, true, {0, Ns});
      { Real *t = ub[2]; ub[2] = ub[1]; ub[1] = ub[0]; ub[0] = t; }
      ring_fill++; steps_done++;
      u0_src = nullptr; u0 = C; u1 = D; bufC = A; bufD = B;
      if (op.timing) { hipEventRecord(ev.second, s); step_ev.push_back(ev); }
      HIPCHK(hipGetLastError());
      if (ring_fill == ring_depth) return flush();
      return PF_OK;
   }

   // ------------------------------------------------------------------------------------------------------------
   void launch_air(hipStream_t s, int xb, int xe) {
      if (xe <= xb) return;
      std::pair<hipEvent_t, hipEvent_t> ev{};
      if (op.timing) {
         if (!ev_pool.empty()) { ev = ev_pool.back(); ev_pool.pop_back(); }
         else { hipEventCreate(&ev.first); hipEventCreate(&ev.second); }
         hipEventRecord(ev.first, s);
      }
      if (lean) launch_air_lean(s, xb, xe);
      else launch_air_march(s, xb, xe);
      if (op.timing) { hipEventRecord(ev.second, s); air_ev.push_back(ev); }
   }

   // x-chunk length of the marching kernels.  op.air_chunk > 0: that many planes; < 0: -air_chunk equal chunks;
   // 0: automatic, from sweeps on MI355X (tools/tune_air.py --chunks=-2,-4,...; 1024^3, 512^3, 256^3, 1/2..1/8 slabs):
   //   lean kernels (lean=true): ~64-plane chunks, but between 512 and 2048 workgroups in total -- fewer starve the
   //   CUs, more only add 2-plane prologues; barrier-free v1 kernels: ~12k workgroups, chunks of >= 6 planes.
   // The split is always even (a short last chunk idles its XCD at the end) and the chunk count even (XCD swizzle).
   int pick_chunk(int nplanes, int64_t tiles, bool lean_rule) const {
      int chunk = op.air_chunk;
      if (chunk < 0) chunk = (int)cdiv(nplanes, std::min<int64_t>(-(int64_t)chunk, nplanes));
      else if (chunk == 0) {
         tiles = std::max<int64_t>(tiles, 1);
         int64_t n;
         if (lean_rule) {
            // slab engines keep >= 1024 WGs: with fewer, every WG is resident for the whole launch and the edge-stream
            // kernels / the RCCL transfer kernel find no free CU slot until the interior launch has finished
            const bool slab = !(op.slab_first && op.slab_last);
            const int64_t lo = cdiv(slab ? 1024 : 512, tiles), hi = std::max(cdiv(2048, tiles), lo);
            n = std::min(std::max<int64_t>(nplanes / 64, lo), hi);
         } else {
            n = std::min<int64_t>(cdiv(256 * 48, tiles), std::max(nplanes / 6, 1));
         }
         if (n > 1 && (n & 1)) n++;
         n = std::max<int64_t>(1, std::min<int64_t>(n, nplanes));
         chunk = (int)cdiv(nplanes, n);
      }
      return std::max(1, std::min(chunk, nplanes));
   }

   // Lanes per row segment of the barrier-free kernels: 64 lanes x 16 B = 1 KiB of z per wave row wastes lanes on narrow
   // grids (Nz=309 -> pitch 320: two 256-column segments, 62 % used).  With 32 or 16 lanes per segment a wave stacks 2 or
   // 4 segments in y instead; pick the width with the least padding (ties: the widest).
   int pick_lw() const {
      constexpr int V = pf::VecOf<Real>::V;
      if (op.debug & 0x300) return (op.debug & 0x100) ? 32 : 16; // tuning override
      if (lw_force) return lw_force;                             // measured at creation (autotune)
      // narrower segments cost extra edge-column loads (two per segment and row; the 13-point kernel needs them on every
      // row of all three planes): worth it only when they save >= 10 % of the padded width (7-point) / 25 % (13-point);
      // measured: Nz=309 7-pt +10 % with 16 lanes, Nz=850 13-pt -4 % with 32 lanes
      const int64_t w64 = cdiv(P, (int64_t)64 * V) * 64 * V;
      const double need = fcc ? 0.75 : 0.90;
      int best = 64;
      int64_t best_w = w64;
      for (int lw : {32, 16}) {
         const int64_t w = cdiv(P, (int64_t)lw * V) * lw * V;
         if (w < best_w && (double)w <= need * (double)w64) { best_w = w; best = lw; }
      }
      return best;
   }
   // tile order of the marching kernels: 2 = XCD-banded inside every x chunk (pf_kernels.h: xcd_band), 1 = one contiguous run
   // of the launch per XCD (round 1), 0 = plain (air_variant | 64); PFFDTD_SWIZZLE overrides for measurements
   // Banded wins on large planes (1024^2: k_air_fcc 2.46 -> 2.32 ms, barrier-free 7-point 2.45 -> 2.29, lean 2.33 -> 2.28;
   // Musikverein 552 x 850: 3.60 -> 3.44), the per-XCD run on small ones, where a whole chunk of planes fits one L2 and a band
   // is a handful of tiles (CTK church 579 x 309, 50 tiles per chunk: 0.392 vs 0.425 ms): banded from 96 tiles per chunk.
   int swizzle_mode(int64_t tiles_per_chunk) const { return order_force >= 0 ? order_force : (tiles_per_chunk >= 96 ? 2 : 1); }
   static uint32_t grid_blocks(int swz, int nzt, int nyt, int nxc) {
      return swz == 2 ? pf::xcd_band_blocks((uint32_t)nzt * nyt, (uint32_t)nxc) : (uint32_t)nzt * nyt * nxc;
   }
   // the barrier-free marching kernels (pf_kernels.h): R = 4 rows per lane, 4 waves stacked in y
   void launch_air_march(hipStream_t s, int xb, int xe) {
      const int lw = (op.debug & 0x400) ? 64 : pick_lw();
      if (lw == 32) launch_march_lw<32>(s, xb, xe);
      else if (lw == 16) launch_march_lw<16>(s, xb, xe);
      else launch_march_lw<64>(s, xb, xe);
   }
   template <int LW> void launch_march_lw(hipStream_t s, int xb, int xe) {
      constexpr int V = pf::VecOf<Real>::V, R = 4, WY = 4, WZ = 1;
      pf::AirParams ap;
      ap.Ny = Ny; ap.P = P; ap.plane = plane;
      ap.x_begin = xb; ap.x_end = xe;
      ap.nzt = (int)cdiv(P, (int64_t)WZ * LW * V);
      ap.nyt = (int)cdiv(Ny - 2, (int64_t)WY * R * (64 / LW));
      const int nplanes = xe - xb;
      const int chunk = pick_chunk(nplanes, (int64_t)ap.nzt * ap.nyt, false);
      ap.chunk = chunk;
      ap.nxc = (int)cdiv(nplanes, chunk);
      ap.swizzle = swizzle_mode((int64_t)ap.nzt * ap.nyt);
      ap.Nx = (int)Nx; ap.Nz = (int)Nz; ap.first = op.slab_first; ap.last = op.slab_last; ap.fold = fold ? 1 : 0;
      ap.swz = swz ? 1 : 0;
      const uint32_t total = grid_blocks(ap.swizzle, ap.nzt, ap.nyt, ap.nxc);
      dim3 g(total), b(64 * WY * WZ);
      if (v1_dst && vg && !fcc) { // autotune: the 7-point kernel writing to a scratch grid
         hipLaunchKernelGGL((pf::k_air_cart<Real, R, WY, WZ, false, true, true, false, LW>), g, b, 0, s, u1, u0, mask, a1, a2, ap, l, v1_dst);
         return;
      }
      if (fcc && abck && u0_src) { // out of place (shell of a temporally blocked pair, creation-time measurement)
         if (sg) hipLaunchKernelGGL((pf::k_air_fcc<Real, R, WY, WZ, true, true, false, true, LW>), g, b, 0, s, u1, u0, mask, a1, a2, ap, l, u0_src, (const int32_t *)nullptr);
         else hipLaunchKernelGGL((pf::k_air_fcc<Real, R, WY, WZ, false, true, false, true, LW>), g, b, 0, s, u1, u0, mask, a1, a2, ap, l, u0_src, (const int32_t *)nullptr);
         return;
      }
#define PF_LAUNCH(K, SG) do { if (vg) hipLaunchKernelGGL((K<Real, R, WY, WZ, SG, true, true, false, LW>), g, b, 0, s, u1, u0, mask, a1, a2, ap, l); \
                              else if (abck) hipLaunchKernelGGL((K<Real, R, WY, WZ, SG, true, false, true, LW>), g, b, 0, s, u1, u0, mask, a1, a2, ap, l); \
                              else hipLaunchKernelGGL((K<Real, R, WY, WZ, SG, true, false, false, LW>), g, b, 0, s, u1, u0, mask, a1, a2, ap, l); } while (0)
      if (fcc) { if (sg) PF_LAUNCH(pf::k_air_fcc, true); else PF_LAUNCH(pf::k_air_fcc, false); }
      else { if (sg) PF_LAUNCH(pf::k_air_cart, true); else PF_LAUNCH(pf::k_air_cart, false); }
#undef PF_LAUNCH
   }

   // the lean fused 7-point kernel (pf_air_fused.h).  R x WY = rows per lane x waves per workgroup; NT = nontemporal u0 traffic
   template <int R, int WY, bool NT = true> void launch_lean_cfg(hipStream_t s, int xb, int xe) {
      pf::LeanParams fp{};
      fp.u1 = u1; fp.u0 = u0; fp.mask = mask;
      fp.plane = plane;
      fp.Nx = (int)Nx; fp.Ny = (int)Ny; fp.Nz = (int)Nz; fp.P = (int)P;
      fp.x_begin = xb; fp.x_end = xe;
      fp.nzt = lean_nzt;
      fp.nyt = (int)cdiv(Ny - 2, (int64_t)WY * R);
      fp.u0_src = u0_src; fp.yt0 = 0; fp.yt_split = -1; fp.yt_hi0 = 0;
      // (row strips of a triple's third step: the strips' tiles reach into the box, whose u^{n+1} -- the step's old value -- is not in
      // memory: those rows are left to k_tb3's own result)
      if (lean_nyt >= 0 && triples()) { fp.skip_y0 = tby0; fp.skip_y1 = tby1; }
      if (lean_nyt >= 0) { // row strips: tiles [0, lean_nyt) and [lean_yt0, all) in units of this configuration's tile height
         const int all = fp.nyt, lo = std::min(lean_nyt, all), hi0 = std::max(std::min(lean_yt0, all), lo);
         fp.yt_split = lo; fp.yt_hi0 = hi0;
         fp.nyt = lo + (all - hi0);
         if (fp.nyt <= 0) return;
      }
      const int nplanes = xe - xb;
      int chunk = pick_chunk(nplanes, (int64_t)fp.nzt * fp.nyt, true);
      fp.chunk = chunk;
      fp.nxc = (int)cdiv(nplanes, chunk);
      if (lean_x2_end > lean_x2_begin) { // a second x slab [lean_x2_begin, lean_x2_end) in the same launch (both thin: one chunk each)
         chunk = std::max(nplanes, lean_x2_end - lean_x2_begin);
         fp.chunk = chunk;
         fp.x_lo_end = xe; fp.x2_begin = lean_x2_begin; fp.x_end = lean_x2_end;
         fp.x2_nlo = 1; fp.nxc = 2;
      }
      fp.swizzle = swizzle_mode((int64_t)fp.nzt * fp.nyt);
      fp.first = op.slab_first; fp.last = op.slab_last;
      fp.do_abc = 1;
      fp.swz = swz ? 1 : 0;
      dim3 g(grid_blocks(fp.swizzle, fp.nzt, fp.nyt, fp.nxc)), b(64 * WY);
      if (sg) hipLaunchKernelGGL((pf::k_air_cart_lean<Real, R, WY, true, NT>), g, b, 0, s, fp, a1, a2, l);
      else hipLaunchKernelGGL((pf::k_air_cart_lean<Real, R, WY, false, NT>), g, b, 0, s, fp, a1, a2, l);
   }
   void launch_air_lean(hipStream_t s, int xb, int xe) {
      // fastest measured on MI355X: fp32 R = 4 x 4 waves, fp64 R = 2 x 8 waves (register budget)
      if (sizeof(Real) == 8) launch_lean_cfg<2, 8>(s, xb, xe);
      else launch_lean_cfg<4, 4>(s, xb, xe);
   }

   void launch_pre(hipStream_t s) {
      if (lean || vg) return; // ghost shell is virtual, u2ba is the old u0 in registers
      launch_flips(s);
      if (Nba && !abck) hipLaunchKernelGGL(pf::k_abc_save<Real>, dim3((unsigned)cdiv(Nba, 256)), dim3(256), 0, s, u0, d_bna, u2ba, Nba);
   }
   void launch_flips(hipStream_t s) {
      dim3 gy((unsigned)cdiv(Nz, 256), (unsigned)Nx);
      if (fold) hipLaunchKernelGGL(pf::k_flip_y<Real>, gy, dim3(256), 0, s, u1, Nx, Ny, P, Nz, 4);
      hipLaunchKernelGGL(pf::k_flip_z<Real>, dim3((unsigned)cdiv(Nx * Ny, 256)), dim3(256), 0, s, u1, Nx * Ny, P, Nz);
      hipLaunchKernelGGL(pf::k_flip_y<Real>, gy, dim3(256), 0, s, u1, Nx, Ny, P, Nz, fold ? 1 : 3);
      if (op.slab_first || op.slab_last)
         hipLaunchKernelGGL(pf::k_flip_x<Real>, dim3((unsigned)cdiv(plane, 256)), dim3(256), 0, s, u1, Nx, plane, op.slab_first, op.slab_last);
   }
   void launch_abc(hipStream_t s, Range r) {
      if (lean || vg || abck || r.e <= r.b) return;
      if (sg) hipLaunchKernelGGL((pf::k_abc_loss<Real, true>), dim3((unsigned)cdiv(r.e - r.b, 256)), dim3(256), 0, s, u0, d_bna, d_Q, u2ba, l, r.b, r.e);
      else hipLaunchKernelGGL((pf::k_abc_loss<Real, false>), dim3((unsigned)cdiv(r.e - r.b, 256)), dim3(256), 0, s, u0, d_bna, d_Q, u2ba, l, r.b, r.e);
   }
   // Virtual-ghost modes: boundary nodes next to the folded ghost row read it from MEMORY, so that one row is kept
   // materialised.  In a split-phase step the main stream only touches planes [1, Nx-1): the slab's ghost planes may
   // be receiving the neighbours' data at that moment (the edge stream, ordered after the exchange, does those).
   int fold_x0 = 0, fold_x1 = 0; // plane range of the next launch_fold_row (set by the step drivers)
   void launch_fold_row(hipStream_t s) {
      if (!((lean || vg) && fold && need_fold_row) || fold_x1 <= fold_x0) return;
      dim3 gy((unsigned)cdiv(Nz, 256), (unsigned)(fold_x1 - fold_x0));
      hipLaunchKernelGGL(pf::k_flip_y<Real>, gy, dim3(256), 0, s, u1 + (int64_t)fold_x0 * plane, (int64_t)(fold_x1 - fold_x0), Ny, P, Nz, 4);
   }
   // rigid + FD in one pass over the boundary list (plane range given on the boundary list)
   void launch_boundary(hipStream_t s, Range r) {
      // (inside step_pair) the branch ODEs of the column strips' lossy nodes ride along in the same launch (k_fd_sel's work)
      // (a launch of its own, k_fd_sel, until round 3: same time within noise, one kernel fewer)
      const bool with_fd = zs_mode == 2 && bnd_sel && bnd_sel == zs_rest && zs_nfd > 0 && r.b == 0 && r.e == zs_nrest;
      if (r.e <= r.b && !with_fd) return;
      launch_fold_row(s);
      const int64_t nfd = with_fd ? zs_nfd : 0;
      dim3 g((unsigned)cdiv(r.e - r.b + nfd, 128)), b(128);
#define PF_BND(F, M) hipLaunchKernelGGL((pf::k_boundary<Real, F, M>), g, b, 0, s, u1, u0, d_bn, d_adj, d_lossy, a2, sl2, P, plane, ub[0], ub[2], d_ssaf, d_mat, d_Mb, d_mq, d_beta, vh1, gh1, bs_vout ? bs_vout : vh1, bs_gout ? bs_gout : gh1, lo2, (int64_t)mb_max, r.b, r.e, u0_src ? u0_src : (const Real *)u0, bnd_sel, swz ? 1 : 0, with_fd ? zs_fd : (const int32_t *)nullptr, nfd, d_bnl, bflags)
      // Each XCD walks runs of 64 consecutive workgroups
      // (8192 nodes, a few node rows) inside a window of 512, so most rows of u^n that consecutive node rows share are asked
      // for by ONE L2: fetched bytes 4.53 -> 3.7 GB on the Musikverein, 0.79-0.81 -> 0.75-0.76 ms (CTK 0.150 -> 0.141).
      // One run per XCD over the whole list fetches least (3.36 GB) and is slowest (0.90 ms: eight places in every stream);
      // profiles/r04_rooms_hbm_traffic.md.  (Box rooms in single steps: 1024^3 384 -> 388 Gvox/s; slabs: the same.)
      // debug 0x100000: plain order; 0x200000: fetch the neighbours inside the wall too.
      const int bnd_g = 64;
      const int bflags = ((!(op.debug & 0x100000)) ? (1 | (bnd_g << 4)) : 0) | ((op.debug & 0x200000) ? 2 : 0);
      if (fcc) { if (sg) PF_BND(true, true); else PF_BND(true, false); }
      else { if (sg) PF_BND(false, true); else PF_BND(false, false); }
#undef PF_BND
   }
   bool boundary_fused() const { return fuse_boundary; }
   void launch_rigid(hipStream_t s, Range r) {
      if (boundary_fused()) { launch_boundary(s, r); return; }
      if (r.e <= r.b) return;
      launch_fold_row(s);
      dim3 g((unsigned)cdiv(r.e - r.b, 256)), b(256);
      if (fcc) {
         if (sg) hipLaunchKernelGGL((pf::k_rigid<Real, true, true>), g, b, 0, s, u1, u0, d_bn, d_adj, a2, sl2, P, plane, r.b, r.e, swz ? 1 : 0);
         else hipLaunchKernelGGL((pf::k_rigid<Real, true, false>), g, b, 0, s, u1, u0, d_bn, d_adj, a2, sl2, P, plane, r.b, r.e, swz ? 1 : 0);
      } else {
         if (sg) hipLaunchKernelGGL((pf::k_rigid<Real, false, true>), g, b, 0, s, u1, u0, d_bn, d_adj, a2, sl2, P, plane, r.b, r.e, swz ? 1 : 0);
         else hipLaunchKernelGGL((pf::k_rigid<Real, false, false>), g, b, 0, s, u1, u0, d_bn, d_adj, a2, sl2, P, plane, r.b, r.e, swz ? 1 : 0);
      }
   }
   void launch_fd(hipStream_t s, Range r) {
      if (boundary_fused()) return; // done by launch_boundary
      if (r.e > r.b)
         hipLaunchKernelGGL(pf::k_fd_boundary<Real>, dim3((unsigned)cdiv(r.e - r.b, 128)), dim3(128), 0, s, u0, d_bnl, ub[0], ub[2], d_ssaf, d_mat, d_Mb, d_mq, d_beta, vh1, gh1, lo2, (int64_t)mb_max, r.b, r.e);
   }
   // receivers on/off + a range of the (sorted) source list
   void launch_io(hipStream_t s, int64_t n, bool receivers, Range src, const int64_t *ctr = nullptr) {
      const int64_t nr = receivers ? Nr : 0;
      const int64_t ns = src.e - src.b;
      if (nr == 0 && ns <= 0) return;
      hipLaunchKernelGGL(pf::k_io<Real>, dim3((unsigned)cdiv(nr + 1, 128)), dim3(128), 0, s, u1, u0, d_out, ring, nr, ring_fill, ring_depth,
                         d_in + src.b, d_insig + src.b * Nt, std::max<int64_t>(ns, 0), Nt, n, ctr);
   }
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

   int wall_streams() { // a slab's wall regions run on two streams of their own (created on first use)
      if (s_wall) return PF_OK;
      int lo_prio = 0, hi_prio = 0;
      hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
      HIPCHK(hipStreamCreateWithPriority(&s_wall, hipStreamNonBlocking, hi_prio));
      HIPCHK(hipStreamCreateWithPriority(&s_wall2, hipStreamNonBlocking, hi_prio));
      HIPCHK(hipEventCreateWithFlags(&ev_wall0, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&ev_wall, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&ev_wall2, hipEventDisableTiming));
      return PF_OK;
   }
   // split-phase step for slab chains: edge planes and every boundary list entry that lives in them first
   // (high-priority stream), interior concurrently on the main stream.
   int step_begin(int64_t n) override {
      if (in_step) return set_err(PF_ERR_STATE, "step_begin called twice");
      if (n < 0 || n >= Nt) return set_err(PF_ERR_ARG, "step %ld outside [0,Nt=%ld)", (long)n, (long)Nt);
      HIPCHK(hipSetDevice(op.device));
      const int xl = 1, xh = (int)Nx - 2;
      // Slab engines with all four grids at hand step in temporally blocked pairs that span two split-phase steps:
      // phase 0 (step n): edge planes n -> n+1 on the edge stream; box n -> n+1, n+2 plus the shell n -> n+1 on the
      // main stream; phase 1 (step n+1): edge planes and shell n+1 -> n+2.  The exchanges in between are the usual ones.
      // Slab engines with FIVE grids and wall regions that fit step in TRIPLES across three split-phase steps (tb3_slab): the edge
      // stream owns three planes per side (single steps, exchanged after every step as always); phase 0 (step n): box n -> n+2, n+3
      // by k_tb3 -- its first and last plane leave their u^{n+1} too --, wall regions n -> n+1, n+2, the planes between edge planes and
      // box, the single-step tiles and the box's own nodes n -> n+1; phase 1: those n+1 -> n+2; phase 2: they and the whole strips
      // beside the box n+2 -> n+3 as one single step, every node of the interior planes by the list kernel.
      if (tb3_slab && (pair_phase > 0 || (n + 2 < Nt && ring_fill + 3 <= ring_depth && xh - xl >= 12))) {
         const int ph = pair_phase;
         if (ph == 0) {
            tb3_pick();
            pA = u0; pB = u1; u0_src = pA; u1 = pB; u0 = bufC;
            wsP[0] = ub[0]; wsP[1] = ub[1]; wsP[2] = ub[2]; bs_vout = vh1b; bs_gout = gh1b;
         }
         fold_x0 = 0; fold_x1 = 0;
         launch_air_lean(s_edge, xl, xl + 3);
         launch_air_lean(s_edge, xh - 2, xh + 1);
         launch_rigid(s_edge, bn_lo3); launch_rigid(s_edge, bn_hi3);
         launch_fd(s_edge, bnl_lo3); launch_fd(s_edge, bnl_hi3);
         launch_io(s_edge, n, false, in_lo3); launch_io(s_edge, n, false, in_hi3);
         HIPCHK(hipEventRecord(ev_edge, s_edge));
         std::pair<hipEvent_t, hipEvent_t> eva{}, evt{};
         auto get_ev = [&]() { std::pair<hipEvent_t, hipEvent_t> e{}; if (!ev_pool.empty()) { e = ev_pool.back(); ev_pool.pop_back(); } else { hipEventCreate(&e.first); hipEventCreate(&e.second); } return e; };
         if (op.timing) { eva = get_ev(); hipEventRecord(eva.first, s_main); }
         if (ph == 0) {
            { int rcs = wall_streams(); if (rcs) return rcs; }
            HIPCHK(hipEventRecord(ev_wall0, s_main));
            HIPCHK(hipStreamWaitEvent(s_wall, ev_wall0, 0));
            HIPCHK(hipStreamWaitEvent(s_wall2, ev_wall0, 0));
            launch_walls(s_wall, s_wall2, pA, pB, bufC, bufD, wsP[0], wsP[1], wsP[2]);
            launch_shell_planes(s_wall, xl + 3, xh - 2, false);
            bnd_sel = wl_rest; launch_rigid(s_wall, {0, wl_nrest}); bnd_sel = nullptr;
            HIPCHK(hipEventRecord(ev_wall, s_wall));
            HIPCHK(hipEventRecord(ev_wall2, s_wall2));
            wall_pending = true;
            if (op.timing) { evt = get_ev(); hipEventRecord(evt.first, s_main); }
            launch_tb3(s_main, pA, pB, bufC, bufD, bufE);
            if (op.timing) { hipEventRecord(evt.second, s_main); tb2_ev.push_back(evt); }
            launch_dirty_tiles(s_main);
            HIPCHK(hipStreamWaitEvent(s_main, ev_wall, 0)); // (a source in those planes is added after their update)
         } else if (ph == 1) {
            launch_shell_planes(s_main, xl + 3, xh - 2);
            bnd_sel = wl_rest; launch_rigid(s_main, {0, wl_nrest}); bnd_sel = nullptr;
         } else {
            launch_shell(s_main, xl + 3, xh - 2);
            launch_rigid(s_main, bn_mid3);
         }
         if (op.timing) { hipEventRecord(eva.second, s_main); air_ev.push_back(eva); }
         launch_fd(s_main, bnl_mid3);
         launch_io(s_main, n, true, in_mid3);
         HIPCHK(hipGetLastError());
         in_step = true;
         pair_now = true; triple_now = true;
         return PF_OK;
      }
      if (tb2_slab && !tb3_slab && (pair_phase == 1 || (n + 1 < Nt && ring_fill + 2 <= ring_depth && xh - xl >= 8))) {
         const bool first_half = pair_phase == 0;
         // with wall regions (init_walls(true)): the first half also steps the row and column strips beside the box TWICE
         // (k_wall2: branch state vh1 -> vh1b, node values P2, P1 -> P0, P1), so every other boundary launch of the pair follows
         // the same buffers: first half state out of place into vh1b and node values into P0, second half both in place (P1)
         if (first_half) {
            pA = u0; pB = u1; u0_src = pA; u1 = pB; u0 = bufC;
            if (wl_on) { wsP[0] = ub[0]; wsP[1] = ub[1]; wsP[2] = ub[2]; bs_vout = vh1b; bs_gout = gh1b; }
         }
         fold_x0 = 0; fold_x1 = 0; // (virtual-ghost modes with a fold row do not block in pairs)
         if (fcc) {
            // 13-point: the ghost shell of u1 lives in memory; its flips touch the whole grid, ghost planes included, so
            // they go on the edge stream (ordered after the exchange that filled those planes) and the interior waits
            launch_flips(s_edge);
            HIPCHK(hipEventRecord(ev_pre, s_edge));
            HIPCHK(hipStreamWaitEvent(s_main, ev_pre, 0));
            launch_air_march(s_edge, xl, xl + 2);   // (k_air_fcc reads u^{n-1} from u0_src)
            launch_air_march(s_edge, xh - 1, xh + 1);
         } else {
            // (the lean kernel explicitly, as launch_shell does: it is the one that honours u0_src -- the barrier-free
            // kernel an engine may have chosen for its single steps reads u^{n-1} from u0, which here is the grid being written)
            launch_air_lean(s_edge, xl, xl + 2);
            launch_air_lean(s_edge, xh - 1, xh + 1);
         }
         launch_rigid(s_edge, bn_lo2); launch_rigid(s_edge, bn_hi2);
         launch_fd(s_edge, bnl_lo2); launch_fd(s_edge, bnl_hi2);
         launch_io(s_edge, n, false, in_lo2); launch_io(s_edge, n, false, in_hi2);
         HIPCHK(hipEventRecord(ev_edge, s_edge));
         std::pair<hipEvent_t, hipEvent_t> eva{}, evt{};
         auto get_ev = [&]() { std::pair<hipEvent_t, hipEvent_t> e{}; if (!ev_pool.empty()) { e = ev_pool.back(); ev_pool.pop_back(); } else { hipEventCreate(&e.first); hipEventCreate(&e.second); } return e; };
         if (op.timing) { eva = get_ev(); hipEventRecord(eva.first, s_main); }
         if (first_half) {
            if (wl_on) { // beside the box kernel, on a stream of their own: a slab's regions are a few hundred waves, each a chain of dependent march steps
               { int rcs = wall_streams(); if (rcs) return rcs; }
               // (the generic blocks -- a 0.3 ms chain of dependent steps at 1/8 of 1024^3 -- on a stream of their own: behind the
               // alike blocks' launches in ONE stream the regions, 0.52 ms, outlasted the box kernel, 0.47)
               HIPCHK(hipEventRecord(ev_wall0, s_main));
               HIPCHK(hipStreamWaitEvent(s_wall, ev_wall0, 0));
               HIPCHK(hipStreamWaitEvent(s_wall2, ev_wall0, 0));
               launch_walls(s_wall, s_wall2, pA, pB, bufC, bufD, wsP[0], wsP[1], wsP[2]);
               // the first step of the planes between the edge planes and the box (an end slab's x wall) and of the boundary nodes
               // no region owns: behind the alike blocks, not behind the box kernel (they only read u^{n-1}, u^n)
               launch_shell_planes(s_wall, xl + 2, xh - 1, false);
               bnd_sel = wl_rest; launch_rigid(s_wall, {0, wl_nrest}); bnd_sel = nullptr;
               HIPCHK(hipEventRecord(ev_wall, s_wall));
               HIPCHK(hipEventRecord(ev_wall2, s_wall2));
               wall_pending = true;
            }
            if (op.timing) { evt = get_ev(); hipEventRecord(evt.first, s_main); }
            launch_tb2(s_main, pA, pB, bufC, bufD);
            if (op.timing) { hipEventRecord(evt.second, s_main); tb2_ev.push_back(evt); }
         }
         if (wl_on && first_half) launch_dirty_tiles(s_main); // (the strips beside the box are the wall regions'; the planes outside it: above)
         else if (wl_on) launch_shell_planes(s_main, xl + 2, xh - 1);
         else launch_shell(s_main, xl + 2, xh - 1);
         if (op.timing) { hipEventRecord(eva.second, s_main); air_ev.push_back(eva); }
         if (wl_on && first_half) HIPCHK(hipStreamWaitEvent(s_main, ev_wall, 0)); // (a source in those planes is added after their update)
         else if (wl_on) { bnd_sel = wl_rest; launch_rigid(s_main, {0, wl_nrest}); bnd_sel = nullptr; }
         else launch_rigid(s_main, bn_mid2);
         launch_fd(s_main, bnl_mid2);
         launch_io(s_main, n, true, in_mid2);
         HIPCHK(hipGetLastError());
         in_step = true;
         pair_now = true;
         return PF_OK;
      }
      if (!(lean || vg)) { // ghost flips / ABC save touch the whole grid: the interior must see them
         launch_pre(s_edge);
         HIPCHK(hipEventRecord(ev_pre, s_edge));
         HIPCHK(hipStreamWaitEvent(s_main, ev_pre, 0));
      }
      // edge stream: first / last owned plane
      fold_x0 = 0; fold_x1 = (int)Nx;
      launch_air(s_edge, xl, xl + 1);
      if (xh > xl) launch_air(s_edge, xh, xh + 1);
      launch_abc(s_edge, bna_lo); launch_abc(s_edge, bna_hi);
      launch_rigid(s_edge, bn_lo); launch_rigid(s_edge, bn_hi);
      launch_fd(s_edge, bnl_lo); launch_fd(s_edge, bnl_hi);
      launch_io(s_edge, n, false, in_lo); launch_io(s_edge, n, false, in_hi);
      HIPCHK(hipEventRecord(ev_edge, s_edge));
      // main stream: interior planes
      fold_x0 = 1; fold_x1 = (int)Nx - 1;
      launch_air(s_main, xl + 1, xh);
      launch_abc(s_main, bna_mid);
      launch_rigid(s_main, bn_mid);
      launch_fd(s_main, bnl_mid);
      launch_io(s_main, n, true, in_mid);
      HIPCHK(hipGetLastError());
      in_step = true;
      return PF_OK;
   }
   int state_grids(void **up, void **uc) override {
      if (in_step || pair_phase) return set_err(PF_ERR_STATE, "pf_engine_state_grids inside a step");
      if (up) *up = u0;
      if (uc) *uc = u1;
      return PF_OK;
   }
   int layout(int64_t *dims, int64_t *pitch, int32_t *exchanged) override {
      if (dims) { dims[0] = Nx; dims[1] = Ny; dims[2] = Nz; }
      if (pitch) *pitch = P;
      if (exchanged) *exchanged = swz ? 1 : 0;
      return PF_OK;
   }
   int halo_ptrs(void **slo, void **shi, void **rlo, void **rhi, size_t *bytes) override {
      // new state is u0 until step_end rotates (gpu_engine.h:1086-1126 sends the same planes)
      if (slo) *slo = u0 + plane;
      if (shi) *shi = u0 + (Nx - 2) * plane;
      if (rlo) *rlo = u0;
      if (rhi) *rhi = u0 + (Nx - 1) * plane;
      if (bytes) *bytes = (size_t)plane * sizeof(Real);
      return PF_OK;
   }
   int step_end(int64_t n) override {
      if (!in_step) return set_err(PF_ERR_STATE, "step_end without step_begin");
      HIPCHK(hipSetDevice(op.device));
      // join.  The next step's edge planes (edge stream) read interior plane 2 / Nx-3: wait for the main stream.
      // The next step's interior (main stream) reads the edge planes but never the ghost planes, so it waits for
      // the edge *compute* only (ev_edge, recorded in step_begin before the exchange was issued) -- the exchange
      // itself stays off the main stream's critical path and only orders the edge stream.
      if (wall_pending) { HIPCHK(hipStreamWaitEvent(s_main, ev_wall, 0)); HIPCHK(hipStreamWaitEvent(s_main, ev_wall2, 0)); wall_pending = false; }
      HIPCHK(hipEventRecord(ev_main, s_main));
      HIPCHK(hipStreamWaitEvent(s_edge, ev_main, 0));
      HIPCHK(hipStreamWaitEvent(s_main, ev_edge, 0));
      in_step = false;
      if (pair_now && triple_now) {
         pair_now = triple_now = false;
         if (pair_phase == 0) {        // u^{n+1} complete in bufC; node values and branch state in place from here on
            ub[0] = ub[2] = wsP[1];
            std::swap(vh1, vh1b); std::swap(gh1, gh1b);
            bs_vout = bs_gout = nullptr;
            u0_src = pB; u1 = bufC; u0 = bufD;
            pair_phase = 1;
         } else if (pair_phase == 1) { // u^{n+2} complete in bufD; node values: u^{n+1} in P0, u^{n+2} in P1 -> u^{n+3} into P2
            ub[0] = wsP[2]; ub[1] = wsP[1]; ub[2] = wsP[0];
            u0_src = bufC; u1 = bufD; u0 = bufE;
            pair_phase = 2;
         } else {                      // triple done: state = (bufD, bufE), the former state grids become the next targets
            ub[0] = wsP[0]; ub[1] = wsP[2]; ub[2] = wsP[1];
            Real *D = bufD, *E = bufE;
            u0_src = nullptr; u0 = D; u1 = E; bufD = pA; bufE = pB;
            pair_phase = 0;
         }
         return after_step(n);
      }
      if (pair_now) {
         pair_now = false;
         if (wl_on) {
            if (pair_phase == 0) { // second half: node values and branch state in place
               ub[0] = ub[2] = wsP[1];
               std::swap(vh1, vh1b); std::swap(gh1, gh1b);
               bs_vout = bs_gout = nullptr;
            } else { ub[0] = wsP[2]; ub[1] = wsP[1]; ub[2] = wsP[0]; }
         } else { Real *t = ub[2]; ub[2] = ub[1]; ub[1] = ub[0]; ub[0] = t; }
         if (pair_phase == 0) { // u^{n+1} is complete in bufC: second half reads u^n as the old state and writes bufD
            u0_src = pB; u1 = bufC; u0 = bufD;
            pair_phase = 1;
         } else {               // pair done: state = (bufC, bufD), the former state grids become the spares
            Real *C = bufC, *D = bufD;
            u0_src = nullptr; u0 = C; u1 = D; bufC = pA; bufD = pB;
            pair_phase = 0;
         }
         return after_step(n);
      }
      rotate();
      return after_step(n);
   }

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
   } else if (sd->real_bytes == 8) {
      auto *e = new Engine<double>();
      rc = e->init(sd, &o);
      impl = e;
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
