// pf_multi.hip -- run_sim across several devices from inside the C call: Z-slab chain along the slowest axis (file Nx),
// one engine per slab, one host thread per slab, one-plane halo exchange per step as peer copies over xGMI.
//
// Replaces the multi-GPU half of the reference's `run_sim` (c_cuda/gpu_engine.h:516-662 split_data, :739-823 index
// localisation, :993-1145 time loop with cudaMemcpyPeerAsync after a full sync), re-thought:
//   * the reference drives every GPU from ONE host thread and exchanges only after all streams have drained
//     (gpu_engine.h:1077-1126 "not async to rest of scheme").  Here every slab has its own host thread (the per-step
//     enqueue cost of a slab, ~65 us, would otherwise serialise: 8 x 65 us > the 0.4 ms a slab of 1024^3/8 takes), the
//     engines' split-phase step (pf_engine_step_begin / _end) computes the edge planes first on a high-priority stream,
//     and each slab PULLS its two ghost planes from its neighbours on that edge stream as soon as the neighbour's edge
//     event fires -- while the interior planes run on the main stream.  One host barrier per step keeps the event
//     bookkeeping race-free (events and plane pointers are double-buffered by step parity).
//   * the lists need not be pre-sorted (the reference refuses unsorted input, gpu_engine.h:688): they are cut by plane
//     range here and each engine sorts its own.
//   * the cut is cost-balanced by default (a wall plane of frequency-dependent nodes costs ~24 interior planes),
//     PF_MULTI_EVEN_SPLIT gives the reference's Nx/G rule (gpu_engine.h:532-550).
// A device id may appear several times in the list ("virtual slabs"): the same code path then runs on one GPU, which is
// how the exchange logic is tested bit for bit on a 1-GPU box (tests/test_hip_multi.py).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pffdtd_hip.h"

extern "C" void pf__set_error(const char *msg); // pf_engine.hip (feeds pf_last_error)

namespace {

struct Slab {
   int64_t x0 = 0, x1 = 0;       // owned global planes [x0, x1)
   int64_t xlo = 0, xhi = 0;     // global planes held locally [xlo, xhi): owned + one ghost plane per interior side
   bool first = false, last = false;
   // host arrays of the local pf_simdata
   std::vector<int64_t> bn, bnl, bna, in, out, out_reorder, out_rows;
   std::vector<uint16_t> adj;
   std::vector<int8_t> K, matl, Q;
   std::vector<uint8_t> ssaf; // Real bytes
   std::vector<double> in_sigs, u_out;
   pf_simdata sd{};
};

int fail(const char *fmt, const char *a = "") {
   char buf[512];
   snprintf(buf, sizeof buf, fmt, a);
   pf__set_error(buf);
   return PF_ERR_ARG;
}

// owned plane ranges.  even: Nx/G planes each, +1 for the first Nx%G (gpu_engine.h:532-550).  balanced: equal estimated
// cost (interior plane = 1; a full plane of lossy nodes with 11 branches = 24, of rigid nodes = 5: measured on MI355X)
int partition(const pf_simdata *sd, int G, bool even, std::vector<int64_t> &cuts) {
   const int64_t Nx = sd->Nx;
   if (G < 1 || G >= Nx) return fail("need 1 <= number of slabs < Nx (gpu_engine.h:682)");
   cuts.assign(G + 1, 0);
   cuts[G] = Nx;
   if (G == 1) return PF_OK;
   if (even) {
      const int64_t base = Nx / G, rem = Nx % G;
      for (int g = 0; g < G; g++) cuts[g + 1] = cuts[g] + base + (g < rem ? 1 : 0);
      return PF_OK;
   }
   const int64_t NzNy = sd->Ny * sd->Nz;
   std::vector<double> nb(Nx, 0.0), nl(Nx, 0.0);
   for (int64_t i = 0; i < sd->Nb; i++) nb[sd->bn_ixyz[i] / NzNy] += 1.0;
   for (int64_t i = 0; i < sd->Nbl; i++) nl[sd->bnl_ixyz[i] / NzNy] += 1.0;
   double mb_scale = 1.0;
   if (sd->Nbl > 0) {
      double s = 0;
      for (int64_t i = 0; i < sd->Nbl; i++) s += (double)sd->Mb[sd->mat_bnl[i]];
      mb_scale = s / (double)sd->Nbl / 11.0;
   }
   std::vector<double> cum(Nx + 1, 0.0);
   for (int64_t x = 0; x < Nx; x++) {
      double c = (x == 0 || x == Nx - 1) ? 0.0 : 1.0; // the global ghost planes are not updated
      c += (24.0 * mb_scale * nl[x] + 5.0 * (nb[x] - nl[x])) / (double)NzNy;
      cum[x + 1] = cum[x] + c;
   }
   for (int g = 1; g < G; g++) {
      const double target = cum[Nx] * (double)g / (double)G;
      int64_t x = (int64_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
      x = std::max(x, cuts[g - 1] + 2);            // every slab updates at least one plane
      x = std::min(x, Nx - 2 * (int64_t)(G - g));
      cuts[g] = x;
   }
   return PF_OK;
}

// local problem of slab g: lists cut to the planes it updates, indices re-based (gpu_engine.h:784-823)
int cut_slab(const pf_simdata *sd, const std::vector<int64_t> &cuts, int g, int G, Slab &s) {
   const int64_t Nx = sd->Nx, NzNy = sd->Ny * sd->Nz, Nt = sd->Nt;
   s.x0 = cuts[g]; s.x1 = cuts[g + 1];
   s.first = g == 0; s.last = g == G - 1;
   s.xlo = s.x0 - (s.first ? 0 : 1);
   s.xhi = s.x1 + (s.last ? 0 : 1);
   const int64_t upd0 = std::max<int64_t>(s.x0, 1), upd1 = std::min<int64_t>(s.x1, Nx - 1);
   if (upd1 - upd0 < 1) return fail("a slab must own at least one interior plane");
   const int64_t off = s.xlo * NzNy, lo = upd0 * NzNy, hi = upd1 * NzNy;
   const int rb = sd->real_bytes;
   for (int64_t i = 0; i < sd->Nb; i++) {
      const int64_t ii = sd->bn_ixyz[i];
      if (ii < lo || ii >= hi) continue;
      s.bn.push_back(ii - off);
      s.adj.push_back(sd->adj_bn[i]);
      if (sd->K_bn) s.K.push_back(sd->K_bn[i]);
   }
   for (int64_t i = 0; i < sd->Nbl; i++) {
      const int64_t ii = sd->bnl_ixyz[i];
      if (ii < lo || ii >= hi) continue;
      s.bnl.push_back(ii - off);
      s.matl.push_back(sd->mat_bnl[i]);
      const uint8_t *p = (const uint8_t *)sd->ssaf_bnl + (size_t)i * rb;
      s.ssaf.insert(s.ssaf.end(), p, p + rb);
   }
   for (int64_t i = 0; i < sd->Nba; i++) {
      const int64_t ii = sd->bna_ixyz[i];
      if (ii < lo || ii >= hi) continue;
      s.bna.push_back(ii - off);
      s.Q.push_back(sd->Q_bna[i]);
   }
   for (int64_t i = 0; i < sd->Ns; i++) {
      const int64_t ii = sd->in_ixyz[i];
      if (ii < lo || ii >= hi) continue;
      s.in.push_back(ii - off);
      s.in_sigs.insert(s.in_sigs.end(), sd->in_sigs + i * Nt, sd->in_sigs + (i + 1) * Nt);
   }
   // receivers read u1 at any owned plane (a global ghost plane included, should someone ask for it)
   for (int64_t i = 0; i < sd->Nr; i++) {
      const int64_t ii = sd->out_ixyz[i];
      if (ii < s.x0 * NzNy || ii >= s.x1 * NzNy) continue;
      s.out.push_back(ii - off);
      s.out_rows.push_back(i);
   }
   s.out_reorder.resize(s.out.size());
   for (size_t i = 0; i < s.out.size(); i++) s.out_reorder[i] = (int64_t)i;
   s.u_out.assign(std::max<size_t>(s.out.size() * (size_t)Nt, 1), 0.0);
   // the ssaf vector must be Real-aligned: std::vector<uint8_t> storage is new[]-aligned (16 B), fine for float/double
   pf_simdata &l = s.sd;
   l = *sd;
   l.Nx = s.xhi - s.xlo;
   l.Npts = l.Nx * NzNy;
   l.bn_ixyz = s.bn.data(); l.adj_bn = s.adj.data(); l.K_bn = sd->K_bn ? s.K.data() : nullptr; l.Nb = (int64_t)s.bn.size();
   l.bnl_ixyz = s.bnl.data(); l.mat_bnl = s.matl.data(); l.ssaf_bnl = s.ssaf.data(); l.Nbl = (int64_t)s.bnl.size();
   l.bna_ixyz = s.bna.data(); l.Q_bna = s.Q.data(); l.Nba = (int64_t)s.bna.size();
   l.in_ixyz = s.in.data(); l.in_sigs = s.in_sigs.data(); l.Ns = (int64_t)s.in.size();
   l.out_ixyz = s.out.data(); l.out_reorder = s.out_reorder.data(); l.Nr = (int64_t)s.out.size();
   l.u_out = s.u_out.data();
   l.bn_mask = nullptr; // every engine rebuilds its own mask from its own boundary nodes (as gpu_engine.h:791)
   return PF_OK;
}

// sense-reversing spin barrier (a step takes 0.3-3 ms; the threads meet within microseconds).  The last thread to arrive
// samples the error flag and publishes it with the release, so that all threads take the same decision to stop.
struct SpinBarrier {
   std::atomic<int> count{0};
   std::atomic<int> sense{0};
   std::atomic<int> stop{0};
   int n = 1;
   bool wait(int &local, const std::atomic<int> &err) {
      local ^= 1;
      if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
         count.store(0, std::memory_order_relaxed);
         stop.store(err.load() != 0 ? 1 : 0, std::memory_order_relaxed);
         sense.store(local, std::memory_order_release);
      } else {
         int spins = 0;
         while (sense.load(std::memory_order_acquire) != local)
            if (++spins > 2000) std::this_thread::yield();
      }
      return stop.load(std::memory_order_relaxed) != 0;
   }
};

struct Shared {
   int G = 1;
   std::vector<Slab> slabs;
   std::vector<int> dev;
   std::vector<pf_engine *> eng;
   std::vector<hipStream_t> edge;
   std::vector<hipEvent_t> ev[2];                                // edge planes of step n computed: [n&1][g]
   std::vector<void *> send_lo[2], send_hi[2], recv_lo[2], recv_hi[2];
   std::vector<void *> grids[4];                                 // caller-owned state grids per slab (pairs need four)
   std::vector<int> paired;
   size_t plane_bytes = 0;
   SpinBarrier bar;
   std::atomic<int> err{0};
   std::string err_msg;
   std::atomic_flag err_lock = ATOMIC_FLAG_INIT;
   pf_opts base{};
   int64_t Nt = 0;
   double t_loop = 0;
   void set_error(int rc, const char *what) {
      int expect = 0;
      if (err.compare_exchange_strong(expect, rc ? rc : PF_ERR_HIP)) {
         while (err_lock.test_and_set()) {}
         err_msg = what;
         err_lock.clear();
      }
   }
};

#define MCHK(g, expr)                                                                                     \
   do {                                                                                                   \
      hipError_t _e = (expr);                                                                             \
      if (_e != hipSuccess) {                                                                             \
         char _b[512];                                                                                    \
         snprintf(_b, sizeof _b, "slab %d: HIP error %s at %s:%d: %s", g, hipGetErrorName(_e), __FILE__, __LINE__, hipGetErrorString(_e)); \
         S.set_error(PF_ERR_HIP, _b);                                                                     \
         return;                                                                                          \
      }                                                                                                   \
   } while (0)
#define ECHK(g, expr)                                                                                     \
   do {                                                                                                   \
      int _rc = (expr);                                                                                   \
      if (_rc != PF_OK) { S.set_error(_rc, pf_last_error()); return; }                                    \
   } while (0)

void create_slab(Shared &S, int g) {
   const Slab &sl = S.slabs[g];
   const int d = S.dev[g];
   MCHK(g, hipSetDevice(d));
   pf_opts o = S.base;
   o.device = d;
   o.slab_first = sl.first; o.slab_last = sl.last;
   o.x_global0 = (int32_t)sl.xlo;
   const size_t gb = pf_grid_bytes(sl.sd.Nx, sl.sd.Ny, sl.sd.Nz, sl.sd.real_bytes);
   // temporally blocked pairs need all four grids in the caller's hands (pf_engine_set_spares); worth it for slabs of
   // >= 96 planes (measured)
   const int flags = S.base.multi_flags;
   const bool want_pairs = !(flags & PF_MULTI_NO_PAIRS) && ((flags & PF_MULTI_FORCE_PAIRS) || sl.sd.Nx - 2 >= 96);
   for (int k = 0; k < 2; k++) {
      void *p = nullptr;
      MCHK(g, hipMalloc(&p, gb));
      MCHK(g, hipMemset(p, 0, gb));
      S.grids[k][g] = p;
   }
   MCHK(g, hipDeviceSynchronize());
   o.ext_u0 = S.grids[0][g]; o.ext_u1 = S.grids[1][g];
   ECHK(g, pf_engine_create(&sl.sd, &o, &S.eng[g]));
   if (want_pairs) {
      // a pool of up to eight grids: the engine keeps the four its pair kernel is fastest on (grid placement)
      std::vector<void *> pool = {S.grids[0][g], S.grids[1][g]};
      int extra = 6;
      if (const char *ev = getenv("PFFDTD_PLACE_EXTRA")) extra = std::min(std::max(atoi(ev), 0), 12) + 2;
      for (int k = 0; k < extra; k++) {
         void *p = nullptr;
         if (hipMalloc(&p, gb) != hipSuccess) { (void)hipGetLastError(); break; } // what fits
         if (hipMemset(p, 0, gb) != hipSuccess) { (void)hipGetLastError(); hipFree(p); break; }
         pool.push_back(p);
      }
      MCHK(g, hipDeviceSynchronize());
      int32_t idx[4] = {0, 1, -1, -1};
      int rc = 0;
      if (pool.size() >= 4) rc = pf_engine_place_grids(S.eng[g], pool.data(), (int32_t)pool.size(), idx);
      if (rc != 0) { for (size_t k = 2; k < pool.size(); k++) hipFree(pool[k]); S.set_error(rc, pf_last_error()); return; }
      for (int k = 0; k < 4; k++) S.grids[k][g] = idx[k] >= 0 ? pool[idx[k]] : nullptr;
      for (size_t k = 0; k < pool.size(); k++)
         if ((int)k != idx[0] && (int)k != idx[1] && (int)k != idx[2] && (int)k != idx[3]) hipFree(pool[k]);
      S.paired[g] = idx[2] >= 0;
   }
   S.edge[g] = (hipStream_t)pf_engine_stream(S.eng[g], 1);
   for (int k = 0; k < 2; k++) MCHK(g, hipEventCreateWithFlags(&S.ev[k][g], hipEventDisableTiming));
   // neighbours' memory: direct peer access where the devices differ (the copies work without it, staged)
   for (int nb : {g - 1, g + 1})
      if (nb >= 0 && nb < S.G && S.dev[nb] != d) {
         int can = 0;
         if (hipDeviceCanAccessPeer(&can, d, S.dev[nb]) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(S.dev[nb], 0);
            if (e != hipSuccess) (void)hipGetLastError(); // already enabled (virtual slabs, repeated runs): fine
         }
      }
}

// ---- the three phases of one step of slab g ----
// A: enqueue the split-phase step, publish the planes to exchange and the event "my edge planes of step n are computed"
void phase_begin(Shared &S, int g, int64_t n) {
   if (S.err.load()) return;
   const int k = (int)(n & 1);
   int rc = pf_engine_step_begin(S.eng[g], n);
   if (rc == PF_OK) rc = pf_engine_halo_ptrs(S.eng[g], &S.send_lo[k][g], &S.send_hi[k][g], &S.recv_lo[k][g], &S.recv_hi[k][g], nullptr);
   if (rc != PF_OK) S.set_error(rc, pf_last_error());
   else if (hipEventRecord(S.ev[k][g], S.edge[g]) != hipSuccess) S.set_error(PF_ERR_HIP, "hipEventRecord failed");
}
// B (after every slab has done A): pull the neighbours' freshly computed edge planes into my ghost planes, on MY edge
// stream: ordered after my own edge kernels of this step (which were the last readers of the grid those ghost planes
// belong to) and after the neighbour's edge event; the interior keeps running on the main stream meanwhile
void phase_pull(Shared &S, int g, int64_t n) {
   const int k = (int)(n & 1), d = S.dev[g];
   auto pull = [&](int nb, void *dst, const void *src) {
      if (hipStreamWaitEvent(S.edge[g], S.ev[k][nb], 0) != hipSuccess) { S.set_error(PF_ERR_HIP, "hipStreamWaitEvent failed"); return; }
      const hipError_t e = (S.dev[nb] == d) ? hipMemcpyAsync(dst, src, S.plane_bytes, hipMemcpyDeviceToDevice, S.edge[g])
                                            : hipMemcpyPeerAsync(dst, d, src, S.dev[nb], S.plane_bytes, S.edge[g]);
      if (e != hipSuccess) S.set_error(PF_ERR_HIP, hipGetErrorString(e));
   };
   if (g > 0) pull(g - 1, S.recv_lo[k][g], S.send_hi[k][g - 1]);         // left neighbour's last owned plane -> my plane 0
   if (g < S.G - 1) pull(g + 1, S.recv_hi[k][g], S.send_lo[k][g + 1]);   // right neighbour's first owned plane -> my last plane
}
// C: join the two streams, rotate the state
void phase_end(Shared &S, int g, int64_t n) {
   const int rc = pf_engine_step_end(S.eng[g], n);
   if (rc != PF_OK) S.set_error(rc, pf_last_error());
}

void finish_slab(Shared &S, int g) {
   if (!S.eng[g]) return;
   hipSetDevice(S.dev[g]);
   if (!S.err.load()) {
      int rc = pf_engine_flush_outputs(S.eng[g]);
      if (rc == PF_OK) rc = pf_engine_sync(S.eng[g]);
      if (rc != PF_OK) S.set_error(rc, pf_last_error());
   }
}

void destroy_slab(Shared &S, int g) {
   hipSetDevice(S.dev[g]);
   if (S.eng[g]) { pf_engine_sync(S.eng[g]); pf_engine_destroy(S.eng[g]); S.eng[g] = nullptr; }
   for (int k = 0; k < 2; k++) if (S.ev[k][g]) hipEventDestroy(S.ev[k][g]);
   for (int k = 0; k < 4; k++) if (S.grids[k][g]) hipFree(S.grids[k][g]);
}

void worker(Shared &S, int g) {
   int local = 0;
   create_slab(S, g);
   bool stop = S.bar.wait(local, S.err); // all engines exist, or everybody leaves
   std::chrono::steady_clock::time_point t0;
   if (g == 0) t0 = std::chrono::steady_clock::now();
   for (int64_t n = 0; n < S.Nt && !stop; n++) {
      hipSetDevice(S.dev[g]);
      phase_begin(S, g, n);
      stop = S.bar.wait(local, S.err); // every slab's edge event of step n is recorded, its plane pointers published
      if (stop) break;
      phase_pull(S, g, n);
      phase_end(S, g, n);
   }
   finish_slab(S, g);
   S.bar.wait(local, S.err);
   if (g == 0) S.t_loop = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

} // namespace

extern "C" {

int pf_slab_partition(const pf_simdata *sd, int32_t nslabs, int32_t even_split, int64_t *cuts) {
   if (!sd || !cuts) return fail("pf_slab_partition: null argument");
   std::vector<int64_t> c;
   const int rc = partition(sd, nslabs, even_split != 0, c);
   if (rc) return rc;
   for (int g = 0; g <= nslabs; g++) cuts[g] = c[g];
   return PF_OK;
}

// run_sim on a chain of slabs, slab g on device devices[g] (ids may repeat).  base: engine options common to all slabs
// (numerics, air_variant, readout_chunk, debug, multi_flags); NULL = defaults.
double pf_run_sim_devices(pf_simdata *sd, int32_t nslabs, const int32_t *devices, const pf_opts *base) {
   if (!sd || nslabs < 1 || !devices) { fail("pf_run_sim_devices: bad argument"); return -1.0; }
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { pf__set_error("no HIP device visible"); return -1.0; }
   for (int g = 0; g < nslabs; g++)
      if (devices[g] < 0 || devices[g] >= ndev) { fail("pf_run_sim_devices: device id out of range"); return -1.0; }
   Shared S;
   if (base) S.base = *base; else pf_opts_default(&S.base);
   if (nslabs == 1) { // plain single-domain engine
      pf_opts o = S.base;
      o.device = devices[0]; o.slab_first = o.slab_last = 1;
      pf_engine *e = nullptr;
      if (pf_engine_create(sd, &o, &e) != PF_OK) return -1.0;
      auto t0 = std::chrono::steady_clock::now();
      int rc = pf_engine_run(e, 0, sd->Nt);
      if (rc == PF_OK) rc = pf_engine_sync(e);
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      std::string keep = pf_last_error();
      pf_engine_destroy(e);
      if (rc != PF_OK) { pf__set_error(keep.c_str()); return -1.0; }
      return el;
   }
   const int G = nslabs;
   std::vector<int64_t> cuts;
   if (partition(sd, G, (S.base.multi_flags & PF_MULTI_EVEN_SPLIT) != 0, cuts)) return -1.0;
   S.G = G; S.Nt = sd->Nt;
   S.slabs.resize(G);
   for (int g = 0; g < G; g++)
      if (cut_slab(sd, cuts, g, G, S.slabs[g])) return -1.0;
   S.dev.assign(devices, devices + G);
   S.eng.assign(G, nullptr);
   S.edge.assign(G, nullptr);
   S.paired.assign(G, 0);
   for (int k = 0; k < 2; k++) {
      S.ev[k].assign(G, nullptr);
      S.send_lo[k].assign(G, nullptr); S.send_hi[k].assign(G, nullptr);
      S.recv_lo[k].assign(G, nullptr); S.recv_hi[k].assign(G, nullptr);
   }
   for (int k = 0; k < 4; k++) S.grids[k].assign(G, nullptr);
   S.plane_bytes = (size_t)(sd->Ny * pf_grid_pitch(sd->Nz, sd->real_bytes)) * (size_t)sd->real_bytes;
   S.bar.n = G;
   if (S.base.multi_flags & PF_MULTI_ONE_THREAD) {
      // the reference's arrangement (one host thread drives every GPU, gpu_engine.h:993-1145): kept for debugging
      for (int g = 0; g < G && !S.err.load(); g++) create_slab(S, g);
      auto t0 = std::chrono::steady_clock::now();
      for (int64_t n = 0; n < S.Nt && !S.err.load(); n++) {
         for (int g = 0; g < G; g++) phase_begin(S, g, n);
         for (int g = 0; g < G && !S.err.load(); g++) { hipSetDevice(S.dev[g]); phase_pull(S, g, n); }
         for (int g = 0; g < G && !S.err.load(); g++) phase_end(S, g, n);
      }
      for (int g = 0; g < G; g++) finish_slab(S, g);
      S.t_loop = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
   } else {
      std::vector<std::thread> th;
      for (int g = 1; g < G; g++) th.emplace_back(worker, std::ref(S), g);
      worker(S, 0);
      for (auto &t : th) t.join();
   }
   // receivers: every slab filled its own rows (gpu_engine.h:1066-1075)
   if (!S.err.load() && sd->u_out)
      for (int g = 0; g < G; g++) {
         const Slab &sl = S.slabs[g];
         for (size_t r = 0; r < sl.out_rows.size(); r++)
            memcpy(sd->u_out + sl.out_rows[r] * sd->Nt, sl.u_out.data() + r * (size_t)sd->Nt, sizeof(double) * (size_t)sd->Nt);
      }
   for (int g = 0; g < G; g++) destroy_slab(S, g);
   if (S.err.load()) { pf__set_error(S.err_msg.c_str()); return -1.0; }
   if (getenv("PFFDTD_VERBOSE")) {
      fprintf(stderr, "pffdtd_hip: %d slabs:", G);
      for (int g = 0; g < G; g++) fprintf(stderr, " [dev %d: planes %ld-%ld%s]", S.dev[g], (long)cuts[g], (long)cuts[g + 1] - 1, S.paired[g] ? ", pairs" : "");
      fprintf(stderr, "\n");
   }
   return S.t_loop;
}

// double run_sim(struct SimData *sd): cpu_engine.h:52 / gpu_engine.h:665.  Like the reference's GPU engine it uses every
// visible device (gpu_engine.h:680-682); PFFDTD_NGPUS=n limits it to the first n, PFFDTD_DEVICES=0,0,1 names the
// chain explicitly (repeats allowed).
double pf_run_sim(pf_simdata *sd) {
   if (!sd) { fail("pf_run_sim: null argument"); return -1.0; }
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { pf__set_error("no HIP device visible"); return -1.0; }
   std::vector<int32_t> devs;
   if (const char *ev = getenv("PFFDTD_DEVICES")) {
      for (const char *p = ev; *p;) {
         char *end = nullptr;
         const long v = strtol(p, &end, 10);
         if (end == p) break;
         devs.push_back((int32_t)v);
         p = (*end == ',') ? end + 1 : end;
      }
   }
   if (devs.empty()) {
      int n = ndev;
      if (const char *ev = getenv("PFFDTD_NGPUS")) n = std::max(1, std::min(atoi(ev), ndev));
      // every slab should keep a few planes of its own (the reference only asks for ngpus < Nx, gpu_engine.h:682)
      n = (int)std::max<int64_t>(1, std::min<int64_t>(n, (sd->Nx - 2) / 4));
      for (int i = 0; i < n; i++) devs.push_back(i);
   }
   return pf_run_sim_devices(sd, (int32_t)devs.size(), devs.data(), nullptr);
}

} // extern "C"
