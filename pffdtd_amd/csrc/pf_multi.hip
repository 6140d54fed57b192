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
//   * two transports for the ghost planes (pf_opts.transport / PFFDTD_TRANSPORT): peer copies PULLED by the receiving slab
//     (hipMemcpyPeerAsync over xGMI; needs hipDeviceCanAccessPeer), or RCCL -- ncclSend / ncclRecv of both planes grouped
//     per slab on its edge stream over one single-process communicator clique (ncclCommInitAll); librccl is loaded at run
//     time, only when that transport is asked for.  The first exchanges of a run are CHECKED: every slab checksums (bit
//     patterns) the planes it sent and received and compares them with its neighbours' (pf_multi_info.exchange_verified).
// A device id may appear several times in the list ("virtual slabs"): the same code path then runs on one GPU, which is
// how the exchange logic is tested bit for bit on a 1-GPU box (tests/test_hip_multi.py).
//
// The chain is an object (pf_multi_create / _run / _destroy) with one persistent host thread per slab, so that a host can
// warm up, time and inspect it (bench.py --gpus N without a process launcher); pf_run_sim_devices is create + run + destroy.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only: the library is dlopen()ed (struct Rccl)
#include <dlfcn.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pffdtd_hip.h"
#include "pf_debug.h" // pf_opts_x: the public options + the development / test switches

extern "C" void pf__set_error(const char *msg); // pf_engine.hip (feeds pf_last_error)
extern "C" int pf__axis_exchange_pays(const pf_simdata *sd, int64_t *counts); // pf_engine.hip

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
// cost (interior plane = 1; a full plane of lossy nodes with 11 branches = 23, of rigid nodes = 5: measured on MI355X -- round 4,
// 1024^3 as 8 ranks with wall regions in the slabs: 132 interior planes 0.310 ms per step; 119 / 114 planes + an x wall, which stays
// single steps, 0.337 / 0.329 -- an end rank is mostly fixed cost, 0.0014 ms per plane against 0.0024 inside)
// along_z: the chain is cut along FILE Z instead (slab engines then store the grid with the x and z axes exchanged: Engine::swz)
// wall_scale: the wall planes' weights (23 / 5 interior planes per full plane of lossy / rigid nodes, a fit at 1024^2 planes, Mb = 11, fp32) times
// this factor -- 1: the constants as they are; pf_multi_create MEASURES the factor on the scene at hand (measure_wall_scale, round 5).
// wall1 (optional): the per-plane wall cost at scale 1, in interior planes.
int partition(const pf_simdata *sd, int G, bool even, std::vector<int64_t> &cuts, bool along_z = false, double wall_scale = 1.0,
              std::vector<double> *wall1 = nullptr) {
   const int64_t Nx = along_z ? sd->Nz : sd->Nx; // planes along the cut axis
   if (G < 1 || G >= Nx) return fail(along_z ? "need 1 <= number of slabs < Nz: this scene's chain is cut along file z (the reference: ngpus < Nx, gpu_engine.h:682)"
                                             : "need 1 <= number of slabs < Nx (gpu_engine.h:682)");
   cuts.assign(G + 1, 0);
   cuts[G] = Nx;
   if (G == 1) return PF_OK;
   if (even) {
      const int64_t base = Nx / G, rem = Nx % G;
      for (int g = 0; g < G; g++) cuts[g + 1] = cuts[g] + base + (g < rem ? 1 : 0);
      return PF_OK;
   }
   const int64_t NzNy = along_z ? sd->Ny * sd->Nx : sd->Ny * sd->Nz; // cells per plane of the cut axis
   std::vector<double> nb(Nx, 0.0), nl(Nx, 0.0);
   auto plane_of = [&](int64_t ii) { return along_z ? ii % sd->Nz : ii / (sd->Ny * sd->Nz); };
   for (int64_t i = 0; i < sd->Nb; i++) nb[plane_of(sd->bn_ixyz[i])] += 1.0;
   for (int64_t i = 0; i < sd->Nbl; i++) nl[plane_of(sd->bnl_ixyz[i])] += 1.0;
   double mb_scale = 1.0;
   if (sd->Nbl > 0) {
      double s = 0;
      for (int64_t i = 0; i < sd->Nbl; i++) s += (double)sd->Mb[sd->mat_bnl[i]];
      mb_scale = s / (double)sd->Nbl / 11.0;
   }
   std::vector<double> cum(Nx + 1, 0.0);
   if (wall1) wall1->assign(Nx, 0.0);
   for (int64_t x = 0; x < Nx; x++) {
      double c = (x == 0 || x == Nx - 1) ? 0.0 : 1.0; // the global ghost planes are not updated
      const double wc = (23.0 * mb_scale * nl[x] + 5.0 * (nb[x] - nl[x])) / (double)NzNy;
      if (wall1) (*wall1)[x] = wc;
      c += wall_scale * wc;
      cum[x + 1] = cum[x] + c;
   }
   // Round 6: a cut keeps CUT_CLEAR planes from every source.  A slab in triples takes its shell's three steps in one pass, recomputing three
   // planes of halo beside its box -- which starts four planes from a cut -- from u^{n-1}, u^n alone: a source there (it is added between the
   // steps) would send the slab back to the two-steps-plus-one shell.  The headline scene's source sits at Nx / 2, exactly where an even
   // number of ranks cuts.
   constexpr int64_t CUT_CLEAR = 8;
   std::vector<int64_t> src_planes;
   for (int64_t i = 0; i < sd->Ns; i++) src_planes.push_back(plane_of(sd->in_ixyz[i]));
   auto clear_of_sources = [&](int64_t x) {
      for (int64_t p : src_planes) if (x > p - CUT_CLEAR && x <= p + CUT_CLEAR) return false; // (planes x-1 | x are the cut's two sides)
      return true;
   };
   auto nudge = [&](int64_t x, int64_t lo, int64_t hi) { // the nearest plane that is clear of every source, if the slab thicknesses allow one
      if (clear_of_sources(x)) return x;
      for (int64_t d = 1; d <= 2 * CUT_CLEAR + 2; d++) {
         if (x - d >= lo && clear_of_sources(x - d)) return x - d;
         if (x + d <= hi && clear_of_sources(x + d)) return x + d;
      }
      return x;
   };
   std::vector<char> pinned(G + 1, 0);
   bool any_pinned = false;
   for (int g = 1; g < G; g++) {
      const double target = cum[Nx] * (double)g / (double)G;
      int64_t x = (int64_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
      const int64_t lo = cuts[g - 1] + 2, hi = Nx - 2 * (int64_t)(G - g);
      x = std::max(x, lo);            // every slab updates at least one plane
      x = std::min(x, hi);
      const int64_t xn = nudge(x, lo, hi);
      if (xn != x) { pinned[g] = 1; any_pinned = true; }
      cuts[g] = xn;
   }
   // A cut that a source pushed aside leaves its two neighbours up to CUT_CLEAR planes apart (1024^3 as 8 ranks, source at Nx / 2: 125 and
   // 142 planes where 133 each was meant -- the thick one was the slowest rank of the chain): such a cut stays where it is and the ranks on
   // either side of it share THEIR part of the cost equally among themselves (four ranks over 504 planes, four over 520).
   if (any_pinned) {
      int a = 0;
      while (a < G) {
         int b = a + 1;
         while (b < G && !pinned[b]) b++;
         const double c0 = cum[cuts[a]], c1 = cum[cuts[b]];
         for (int g = a + 1; g < b; g++) {
            const double target = c0 + (c1 - c0) * (double)(g - a) / (double)(b - a);
            int64_t x = (int64_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
            const int64_t lo = cuts[g - 1] + 2, hi = cuts[b] - 2 * (int64_t)(b - g);
            x = std::min(std::max(x, lo), hi);
            cuts[g] = nudge(x, lo, hi);
         }
         a = b;
      }
   }
   return PF_OK;
}

// the same cut along FILE Z: slab g holds the file's columns z in [xlo, xhi) of every row; its local file has Nz = xhi - xlo
int cut_slab_z(const pf_simdata *sd, int64_t upd0, int64_t upd1, Slab &s) {
   const int64_t Nz = sd->Nz, Nt = sd->Nt, nzl = s.xhi - s.xlo;
   const int rb = sd->real_bytes;
   auto in_upd = [&](int64_t ii) { const int64_t z = ii % Nz; return z >= upd0 && z < upd1; };
   auto local = [&](int64_t ii) { return (ii / Nz) * nzl + (ii % Nz - s.xlo); };
   for (int64_t i = 0; i < sd->Nb; i++) {
      const int64_t ii = sd->bn_ixyz[i];
      if (!in_upd(ii)) continue;
      s.bn.push_back(local(ii));
      s.adj.push_back(sd->adj_bn[i]);
      if (sd->K_bn) s.K.push_back(sd->K_bn[i]);
   }
   for (int64_t i = 0; i < sd->Nbl; i++) {
      const int64_t ii = sd->bnl_ixyz[i];
      if (!in_upd(ii)) continue;
      s.bnl.push_back(local(ii));
      s.matl.push_back(sd->mat_bnl[i]);
      const uint8_t *p = (const uint8_t *)sd->ssaf_bnl + (size_t)i * rb;
      s.ssaf.insert(s.ssaf.end(), p, p + rb);
   }
   for (int64_t i = 0; i < sd->Nba; i++) {
      const int64_t ii = sd->bna_ixyz[i];
      if (!in_upd(ii)) continue;
      s.bna.push_back(local(ii));
      s.Q.push_back(sd->Q_bna[i]);
   }
   for (int64_t i = 0; i < sd->Ns; i++) {
      const int64_t ii = sd->in_ixyz[i];
      if (!in_upd(ii)) continue;
      s.in.push_back(local(ii));
      s.in_sigs.insert(s.in_sigs.end(), sd->in_sigs + i * Nt, sd->in_sigs + (i + 1) * Nt);
   }
   for (int64_t i = 0; i < sd->Nr; i++) { // receivers read u1 at any owned column (a global ghost column included)
      const int64_t ii = sd->out_ixyz[i], z = ii % Nz;
      if (z < s.x0 || z >= s.x1) continue;
      s.out.push_back(local(ii));
      s.out_rows.push_back(i);
   }
   s.out_reorder.resize(s.out.size());
   for (size_t i = 0; i < s.out.size(); i++) s.out_reorder[i] = (int64_t)i;
   s.u_out.assign(std::max<size_t>(s.out.size() * (size_t)Nt, 1), 0.0);
   pf_simdata &l = s.sd;
   l = *sd;
   l.Nz = nzl;
   l.Npts = l.Nx * l.Ny * nzl;
   l.bn_ixyz = s.bn.data(); l.adj_bn = s.adj.data(); l.K_bn = sd->K_bn ? s.K.data() : nullptr; l.Nb = (int64_t)s.bn.size();
   l.bnl_ixyz = s.bnl.data(); l.mat_bnl = s.matl.data(); l.ssaf_bnl = s.ssaf.data(); l.Nbl = (int64_t)s.bnl.size();
   l.bna_ixyz = s.bna.data(); l.Q_bna = s.Q.data(); l.Nba = (int64_t)s.bna.size();
   l.in_ixyz = s.in.data(); l.in_sigs = s.in_sigs.data(); l.Ns = (int64_t)s.in.size();
   l.out_ixyz = s.out.data(); l.out_reorder = s.out_reorder.data(); l.Nr = (int64_t)s.out.size();
   l.u_out = s.u_out.data();
   l.bn_mask = nullptr;
   return PF_OK;
}

// local problem of slab g: lists cut to the planes it updates, indices re-based (gpu_engine.h:784-823)
int cut_slab(const pf_simdata *sd, const std::vector<int64_t> &cuts, int g, int G, Slab &s, bool along_z = false) {
   const int64_t Nx = along_z ? sd->Nz : sd->Nx, NzNy = sd->Ny * sd->Nz, Nt = sd->Nt;
   s.x0 = cuts[g]; s.x1 = cuts[g + 1];
   s.first = g == 0; s.last = g == G - 1;
   s.xlo = s.x0 - (s.first ? 0 : 1);
   s.xhi = s.x1 + (s.last ? 0 : 1);
   const int64_t upd0 = std::max<int64_t>(s.x0, 1), upd1 = std::min<int64_t>(s.x1, Nx - 1);
   if (upd1 - upd0 < 1) return fail("a slab must own at least one interior plane");
   if (along_z) return cut_slab_z(sd, upd0, upd1, s);
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

// ---- RCCL, loaded on demand (ncclSend / ncclRecv over xGMI as the second transport; gpu_engine.h:1086-1126 is the
// reference's peer-copy counterpart).  The library is looked up (1) among what the process already has (a Python host that
// imported torch carries torch's own librccl), (2) next to the HIP runtime in use, (3) on the loader's path.
struct Rccl {
   void *h = nullptr;
   decltype(&ncclCommInitAll) CommInitAll = nullptr;
   decltype(&ncclCommDestroy) CommDestroy = nullptr;
   decltype(&ncclGroupStart) GroupStart = nullptr;
   decltype(&ncclGroupEnd) GroupEnd = nullptr;
   decltype(&ncclSend) Send = nullptr;
   decltype(&ncclRecv) Recv = nullptr;
   decltype(&ncclGetErrorString) GetErrorString = nullptr;
   decltype(&ncclGetVersion) GetVersion = nullptr;
   decltype(&ncclGetUniqueId) GetUniqueId = nullptr;   // (these two: the one-process-per-device exchange, pf_rccl_*)
   decltype(&ncclCommInitRank) CommInitRank = nullptr;
   std::string where;
   bool load(std::string &err) {
      if (h) return true;
      std::vector<std::string> cand;
      if (const char *ev = getenv("PFFDTD_RCCL_LIB")) cand.push_back(ev);
      Dl_info di{};
      if (dladdr((void *)&hipGetDeviceCount, &di) && di.dli_fname) {
         std::string dir = di.dli_fname;
         const size_t k = dir.rfind('/');
         if (k != std::string::npos) { dir.resize(k); cand.push_back(dir + "/librccl.so.1"); cand.push_back(dir + "/librccl.so"); }
      }
      cand.push_back("librccl.so.1");
      cand.push_back("librccl.so");
      void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
      if (lib) where = "librccl.so.1 (already loaded)";
      std::string last_err; // (dlerror() hands its message out once and clears it: read it right after the failing call)
      for (size_t i = 0; i < cand.size() && !lib; i++) {
         lib = dlopen(cand[i].c_str(), RTLD_NOW | RTLD_LOCAL);
         if (lib) where = cand[i];
         else if (const char *de = dlerror()) last_err = de;
      }
      if (!lib) { err = std::string("librccl not found (") + (last_err.empty() ? "dlopen failed" : last_err.c_str()) + "); PFFDTD_RCCL_LIB names it explicitly"; return false; }
#define PF_SYM(field, name) field = (decltype(field))dlsym(lib, name); if (!field) { err = std::string("librccl: symbol ") + name + " missing"; dlclose(lib); return false; }
      PF_SYM(CommInitAll, "ncclCommInitAll") PF_SYM(CommDestroy, "ncclCommDestroy") PF_SYM(GroupStart, "ncclGroupStart")
      PF_SYM(GroupEnd, "ncclGroupEnd") PF_SYM(Send, "ncclSend") PF_SYM(Recv, "ncclRecv")
      PF_SYM(GetErrorString, "ncclGetErrorString") PF_SYM(GetVersion, "ncclGetVersion")
      PF_SYM(GetUniqueId, "ncclGetUniqueId") PF_SYM(CommInitRank, "ncclCommInitRank")
#undef PF_SYM
      h = lib;
      return true;
   }
};
Rccl g_rccl;
std::mutex g_rccl_mu;

// sense-reversing spin barrier (a step takes 0.3-3 ms; the threads meet within microseconds).  The last thread to arrive
// samples the error flag and publishes it with the release, so that all threads take the same decision to stop.
// WATCHDOG: a thread that has waited longer than timeout_s gives up -- a neighbour's thread is stuck inside a driver or RCCL call
// (a hung device, a collective that never completes) -- raises `timed_out` and returns "stop"; every other waiting thread does
// the same within its own timeout, so a hang becomes an error (pf_last_error) instead of a process that never returns.  The
// barrier is unusable afterwards (the chain is in its error state and only waits to be destroyed).
struct SpinBarrier {
   std::atomic<int> count{0};
   std::atomic<int> sense{0};
   std::atomic<int> stop{0};
   std::atomic<int> timed_out{0};
   int n = 1;
   bool wait(int &local, const std::atomic<int> &err, double timeout_s = 0.0) {
      if (timed_out.load(std::memory_order_relaxed)) return true;
      local ^= 1;
      if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
         count.store(0, std::memory_order_relaxed);
         stop.store(err.load() != 0 ? 1 : 0, std::memory_order_relaxed);
         sense.store(local, std::memory_order_release);
      } else {
         int64_t spins = 0;
         bool timing = false;
         std::chrono::steady_clock::time_point t0;
         while (sense.load(std::memory_order_acquire) != local) {
            if (++spins <= 2000) continue;
            std::this_thread::yield();
            if (timeout_s > 0 && (spins & 255) == 0) {
               if (timed_out.load(std::memory_order_relaxed)) return true;
               const auto now = std::chrono::steady_clock::now();
               if (!timing) { timing = true; t0 = now; }
               else if (std::chrono::duration<double>(now - t0).count() > timeout_s) { timed_out.store(1); return true; }
            }
         }
      }
      return stop.load(std::memory_order_relaxed) != 0 || timed_out.load(std::memory_order_relaxed) != 0;
   }
};
// seconds a slab thread waits for the others before it declares the chain hung (PFFDTD_BARRIER_TIMEOUT_S; creation: ten times that)
double barrier_timeout() {
   if (const char *ev = getenv("PFFDTD_BARRIER_TIMEOUT_S")) { const double v = atof(ev); if (v > 0) return v; }
   return 120.0;
}

enum { TR_PEER = PF_TRANSPORT_PEER, TR_RCCL = PF_TRANSPORT_RCCL, TR_HOST = PF_TRANSPORT_HOST };

struct Shared {
   int G = 1;
   int only = -1;                                                // >= 0: the one slab that exists (pf_opts.only_slab: cost model of a rank)
   std::vector<Slab> slabs;
   std::vector<int64_t> cuts;
   std::vector<int> dev;
   std::vector<pf_engine *> eng;
   std::vector<hipStream_t> edge;
   std::vector<hipEvent_t> ev[2];                                // edge planes of step n computed: [n&1][g]
   std::vector<void *> send_lo[2], send_hi[2], recv_lo[2], recv_hi[2];
   std::vector<void *> grids[5];                                 // caller-owned state grids per slab (pairs need four, triples five)
   std::vector<int> paired;
   size_t plane_bytes = 0;
   SpinBarrier bar;
   std::atomic<int> err{0};
   std::string err_msg;
   std::atomic_flag err_lock = ATOMIC_FLAG_INIT;
   pf_opts_x base{};
   int64_t Nt = 0;
   double t_loop = 0;
   // transport
   bool along_z = false;                                         // the chain is cut along file z, its engines store the axes exchanged
   int transport = TR_PEER;
   bool rccl_self = false;                                       // every slab on one device: one 1-rank communicator per slab,
                                                                 // planes sent to itself (the RCCL code path on a 1-GPU box)
   std::vector<ncclComm_t> comm;                                 // [g]
   std::vector<int> rank;                                        // [g] peer rank of slab g in the clique
   // host-staged transport (the last resort: neither peer access nor a working RCCL): slab g copies its two edge planes into a
   // pinned bounce buffer of its own on its edge stream, its neighbours copy them out on theirs; the two sides meet on the HOST
   // (hipEventSynchronize), nothing device-to-device is asked of the driver
   std::vector<void *> hstage[2];                                // [n&1][g]: lo plane, hi plane (2 * plane_bytes, pinned, portable)
   std::vector<hipEvent_t> ev_d2h[2], ev_h2d[2];                 // [n&1][g]: my planes are in my buffer / my ghost planes have left the neighbours' buffers
   std::string transport_note;                                   // why this transport (fallbacks taken)
   double bar_timeout = 120.0;
   int faults = 0;                                               // the test switch `test_faults` (csrc/pf_debug.h)
   double wall_scale = 1.0;                                      // factor on the wall planes' weights the chain was cut with
   bool wall_measured = false;                                   // ... measured at creation (pf_slab_wall_scale)
   // exchange self-check: the first `verify_n` exchanges after creation
   int64_t verify_n = 0;
   int64_t drop_step = -1; // test hook (test switch `test_drop_exchange`): slab 1 misses the planes of that step; the self-check then always covers it
   std::vector<int64_t> steps_done;                              // [g]
   std::vector<uint64_t> sums;                                   // [g*4 + {send_lo, send_hi, recv_lo, recv_hi}]
   std::vector<std::vector<uint8_t>> hbuf;                       // [g] host staging for the checksums
   std::atomic<int> verify_bad{0}, verify_nonzero{0};
   std::atomic<int64_t> verify_checked{0};
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
#define NCHK(g, expr)                                                                                     \
   do {                                                                                                   \
      ncclResult_t _r = (expr);                                                                           \
      if (_r != ncclSuccess) {                                                                            \
         char _b[512];                                                                                    \
         snprintf(_b, sizeof _b, "slab %d: RCCL error at %s:%d: %s", g, __FILE__, __LINE__, g_rccl.GetErrorString(_r)); \
         S.set_error(PF_ERR_HIP, _b);                                                                     \
         return;                                                                                          \
      }                                                                                                   \
   } while (0)

// Slabs that share a physical device create their engines one after the other: the creation-time measurements of one
// must not run beside another's, and an engine's optional allocations must not starve a neighbour's mandatory ones.
std::mutex g_dev_mu[64];

void create_slab(Shared &S, int g) {
   const Slab &sl = S.slabs[g];
   const int d = S.dev[g];
   std::lock_guard<std::mutex> dev_lock(g_dev_mu[d & 63]);
   MCHK(g, hipSetDevice(d));
   pf_opts_x o = S.base;
   o.device = d;
   o.slab_first = sl.first; o.slab_last = sl.last;
   o.x_global0 = (int32_t)sl.xlo;
   // (cut along file z: the engines store planes of file z, Ny rows of pitch(Nx) each -- debug 0x1000 -- and step singly)
   const size_t gb = S.along_z ? pf_grid_bytes(sl.sd.Nz, sl.sd.Ny, sl.sd.Nx, sl.sd.real_bytes) : pf_grid_bytes(sl.sd.Nx, sl.sd.Ny, sl.sd.Nz, sl.sd.real_bytes);
   if (S.along_z) o.layout = PF_LAYOUT_EXCHANGED;
   // temporally blocked pairs need all four grids in the caller's hands (pf_engine_set_spares); worth it for slabs of
   // >= 96 planes (measured)
   const int flags = S.base.multi_flags;
   // (a chain cut along file z stores its grids with the axes exchanged, which steps singly: no pool, no placement search)
   const bool want_pairs = !S.along_z && !(flags & PF_MULTI_NO_PAIRS) && ((flags & PF_MULTI_FORCE_PAIRS) || sl.sd.Nx - 2 >= 96);
   for (int k = 0; k < 2; k++) {
      void *p = nullptr;
      MCHK(g, hipMalloc(&p, gb));
      MCHK(g, hipMemset(p, 0, gb));
      S.grids[k][g] = p;
   }
   MCHK(g, hipDeviceSynchronize());
   o.ext_u0 = S.grids[0][g]; o.ext_u1 = S.grids[1][g];
   ECHK(g, pf__engine_create_x(&sl.sd, &o, &S.eng[g]));
   if (want_pairs) {
      // a pool of up to eight grids: the engine keeps the four its pair kernel is fastest on (grid placement).
      // The pool is bounded by what the device has free, less a reserve (two grids or 1/16 of the device, whichever is
      // larger) for the neighbours' engines: slabs that share a device are created one at a time (g_dev_mu) and the
      // rejected candidates are freed before the next one starts.
      std::vector<void *> pool = {S.grids[0][g], S.grids[1][g]};
      int extra = 6;
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
         const size_t reserve = std::max(2 * gb, total_b / 16);
         extra = (int)std::min<size_t>((size_t)extra, free_b > reserve ? (free_b - reserve) / gb : 0);
      }
      for (int k = 0; k < extra; k++) {
         void *p = nullptr;
         if (hipMalloc(&p, gb) != hipSuccess) { (void)hipGetLastError(); break; } // what fits
         if (hipMemset(p, 0, gb) != hipSuccess) { (void)hipGetLastError(); hipFree(p); break; }
         pool.push_back(p);
      }
      MCHK(g, hipDeviceSynchronize());
      int32_t idx[5] = {0, 1, -1, -1, -1};
      int rc = 0;
      // (five or more grids on offer: the engine may step in triples, pf_engine_place_grids5; PF_MULTI_NO_TRIPLES: pairs at most)
      if (pool.size() >= 5 && !(flags & PF_MULTI_NO_TRIPLES)) rc = pf_engine_place_grids5(S.eng[g], pool.data(), (int32_t)pool.size(), idx);
      else if (pool.size() >= 4) rc = pf_engine_place_grids(S.eng[g], pool.data(), (int32_t)pool.size(), idx);
      if (rc != 0) { for (size_t k = 2; k < pool.size(); k++) hipFree(pool[k]); S.set_error(rc, pf_last_error()); return; }
      for (int k = 0; k < 5; k++) S.grids[k][g] = idx[k] >= 0 ? pool[idx[k]] : nullptr;
      for (size_t k = 0; k < pool.size(); k++)
         if ((int)k != idx[0] && (int)k != idx[1] && (int)k != idx[2] && (int)k != idx[3] && (int)k != idx[4]) hipFree(pool[k]);
      S.paired[g] = idx[4] >= 0 ? 3 : (idx[2] >= 0 ? 1 : 0); // (steps per pass: 3 triples, 1 = pairs, 0 single steps)
   }
   S.edge[g] = (hipStream_t)pf_engine_stream(S.eng[g], 1);
   for (int k = 0; k < 2; k++) MCHK(g, hipEventCreateWithFlags(&S.ev[k][g], hipEventDisableTiming));
   // neighbours' memory: direct peer access (checked in choose_transport) for the pulled copies
   if (S.transport == TR_PEER)
      for (int nb : {g - 1, g + 1})
         if (nb >= 0 && nb < S.G && S.dev[nb] != d) {
            const hipError_t e = hipDeviceEnablePeerAccess(S.dev[nb], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
               char b[256];
               snprintf(b, sizeof b, "slab %d: hipDeviceEnablePeerAccess(%d) from device %d failed: %s", g, S.dev[nb], d, hipGetErrorString(e));
               S.set_error(PF_ERR_HIP, b);
               return;
            }
            (void)hipGetLastError();
         }
   if (S.transport == TR_HOST)
      for (int k = 0; k < 2; k++) {
         MCHK(g, hipHostMalloc(&S.hstage[k][g], 2 * S.plane_bytes, hipHostMallocPortable));
         MCHK(g, hipEventCreateWithFlags(&S.ev_d2h[k][g], hipEventDisableTiming));
         MCHK(g, hipEventCreateWithFlags(&S.ev_h2d[k][g], hipEventDisableTiming));
      }
   if (S.verify_n > 0) S.hbuf[g].resize(4 * S.plane_bytes);
}

// Which transport carries the ghost planes?  requested: PF_TRANSPORT_AUTO / _PEER / _RCCL / _HOST (pf_opts.transport), overridden by
// PFFDTD_TRANSPORT=peer|rccl|host|auto.  AUTO = peer copies when every neighbouring pair of devices can access each other's
// memory (hipDeviceCanAccessPeer), else RCCL, else -- librccl missing, its communicators failing or not coming back within
// PFFDTD_RCCL_INIT_TIMEOUT_S (60) seconds -- host-staged copies through pinned bounce buffers: slower, but it needs nothing from
// the driver beyond device <-> pinned-host copies, so a first contact with a new multi-GPU box cannot end without a run.  An
// EXPLICITLY requested transport that is not available is an error, never silently replaced.
int init_rccl(Shared &S, bool all_same, std::string &why) {
   std::lock_guard<std::mutex> lk(g_rccl_mu);
   if (S.faults & 2) { why = "RCCL switched off by the test switch `test_faults` (csrc/pf_debug.h)"; return PF_ERR_ARG; }
   if (!g_rccl.load(why)) return PF_ERR_ARG;
   S.comm.assign(S.G, nullptr);
   S.rank.assign(S.G, 0);
   S.rccl_self = all_same && S.G > 1;
   // RCCL greets with a version banner on STDOUT when NCCL_DEBUG asks for one; a host that prints machine-readable results
   // there (bench.py: one JSON line) must not find it in between: stdout points at stderr while the communicators are made
   // (process-wide and not thread-safe: another thread of the host writing to stdout in this window lands on stderr; chains
   // are created from one thread, before the time loop)
   fflush(stdout);
   const int saved_out = dup(1);
   if (saved_out >= 0) dup2(2, 1);
   // the communicators are made on a helper thread: a rendezvous that never completes (seen on mis-configured boxes) must not
   // hang the caller.  After the timeout the helper is abandoned (it holds only heap state of its own).
   struct Job { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t r = ncclSuccess; std::vector<ncclComm_t> comm; std::vector<int> dev; bool self = false; int only = -1; };
   auto job = std::make_shared<Job>();
   job->comm.assign(S.G, nullptr); job->dev = S.dev; job->self = S.rccl_self; job->only = S.only;
   std::thread([job] {
      ncclResult_t r = ncclSuccess;
      const int G = (int)job->dev.size();
      if (job->self) {
         // virtual slabs: slab g's communicator has ONE rank (the device); its exchange sends the neighbour's plane -- same
         // device, directly addressable -- to itself.  Group semantics, stream ordering and error paths as on a real chain.
         for (int g = 0; g < G && r == ncclSuccess; g++) { if (job->only >= 0 && g != job->only) continue; const int d = job->dev[g]; r = g_rccl.CommInitAll(&job->comm[g], 1, &d); }
      } else r = g_rccl.CommInitAll(job->comm.data(), G, job->dev.data()); // one clique over the chain's devices, rank g = slab g
      { std::lock_guard<std::mutex> l2(job->mu); job->r = r; job->done = true; }
      job->cv.notify_all();
   }).detach();
   double tmo = 60.0;
   if (const char *ev = getenv("PFFDTD_RCCL_INIT_TIMEOUT_S")) { const double v = atof(ev); if (v > 0) tmo = v; }
   bool finished;
   { std::unique_lock<std::mutex> l2(job->mu); finished = job->cv.wait_for(l2, std::chrono::duration<double>(tmo), [&] { return job->done; }); }
   if (saved_out >= 0) { fflush(stdout); dup2(saved_out, 1); close(saved_out); }
   if (!finished) { char b[160]; snprintf(b, sizeof b, "ncclCommInitAll over %d device(s) did not return within %.0f s", S.rccl_self ? 1 : S.G, tmo); why = b; S.comm.clear(); return PF_ERR_HIP; }
   if (job->r != ncclSuccess) {
      char b[384];
      snprintf(b, sizeof b, "ncclCommInitAll over %d device(s) failed: %s", S.rccl_self ? 1 : S.G, g_rccl.GetErrorString(job->r));
      why = b;
      for (auto &c : job->comm) if (c) { g_rccl.CommDestroy(c); c = nullptr; }
      S.comm.clear();
      return PF_ERR_HIP;
   }
   S.comm = job->comm;
   if (!S.rccl_self) for (int g = 0; g < S.G; g++) S.rank[g] = g;
   return PF_OK;
}
int choose_transport(Shared &S, int requested) {
   if (const char *ev = getenv("PFFDTD_TRANSPORT")) {
      if (!strcmp(ev, "peer")) requested = PF_TRANSPORT_PEER;
      else if (!strcmp(ev, "rccl")) requested = PF_TRANSPORT_RCCL;
      else if (!strcmp(ev, "host")) requested = PF_TRANSPORT_HOST;
      else if (!strcmp(ev, "auto") || !*ev) requested = PF_TRANSPORT_AUTO;
      else return fail("PFFDTD_TRANSPORT must be peer, rccl, host or auto (got '%s')", ev);
   }
   if (requested != PF_TRANSPORT_AUTO && requested != PF_TRANSPORT_PEER && requested != PF_TRANSPORT_RCCL && requested != PF_TRANSPORT_HOST)
      return fail("pf_opts.transport must be PF_TRANSPORT_AUTO, _PEER, _RCCL or _HOST");
   bool all_same = true, all_distinct = true, peer_ok = true;
   int bad_a = -1, bad_b = -1;
   for (int g = 0; g < S.G; g++) {
      for (int k = 0; k < g; k++) if (S.dev[k] == S.dev[g]) all_distinct = false;
      if (S.dev[g] != S.dev[0]) all_same = false;
      if (g + 1 < S.G && S.dev[g] != S.dev[g + 1]) {
         int ab = 0, ba = 0;
         if (hipDeviceCanAccessPeer(&ab, S.dev[g], S.dev[g + 1]) != hipSuccess) ab = 0;
         if (hipDeviceCanAccessPeer(&ba, S.dev[g + 1], S.dev[g]) != hipSuccess) ba = 0;
         (void)hipGetLastError();
         if (!(ab && ba) && peer_ok) { peer_ok = false; bad_a = S.dev[g]; bad_b = S.dev[g + 1]; }
      }
   }
   if ((S.faults & 1) && peer_ok) { peer_ok = false; bad_a = S.dev[0]; bad_b = S.dev[S.G > 1 ? 1 : 0]; } // (test hook: pretend there is no peer access)
   char nopeer[200] = "";
   if (!peer_ok) snprintf(nopeer, sizeof nopeer, "devices %d and %d cannot access each other's memory (hipDeviceCanAccessPeer)", bad_a, bad_b);
   if (requested == PF_TRANSPORT_PEER && !peer_ok) {
      char b[400];
      snprintf(b, sizeof b, "%s: the peer-copy transport would stage through the host; use the RCCL or the host-staged transport "
                            "(pf_opts.transport / PFFDTD_TRANSPORT=rccl|host) or leave the choice to the library (auto)", nopeer);
      pf__set_error(b);
      return PF_ERR_ARG;
   }
   if (requested == PF_TRANSPORT_HOST) { S.transport = TR_HOST; S.transport_note = "requested"; return PF_OK; }
   if (requested == PF_TRANSPORT_PEER || (requested == PF_TRANSPORT_AUTO && peer_ok)) { S.transport = TR_PEER; return PF_OK; }
   // RCCL: asked for, or AUTO without peer access
   std::string why;
   int rc = PF_OK;
   if (!(all_distinct || all_same)) { why = "the RCCL transport needs every slab on its own device (or all slabs on ONE device: self-communicators, tests)"; rc = PF_ERR_ARG; }
   else rc = init_rccl(S, all_same, why);
   if (rc == PF_OK) { S.transport = TR_RCCL; if (requested == PF_TRANSPORT_AUTO) S.transport_note = nopeer; return PF_OK; }
   if (requested == PF_TRANSPORT_RCCL) { pf__set_error(why.c_str()); return rc; }
   S.transport = TR_HOST; // the last resort
   S.rccl_self = false;
   S.transport_note = std::string(nopeer) + "; RCCL: " + why;
   fprintf(stderr, "pffdtd_hip: ghost planes travel HOST-STAGED (pinned bounce buffers): %s\n", S.transport_note.c_str());
   return PF_OK;
}

// ---- the phases of one step of slab g ----
// A: enqueue the split-phase step, publish the planes to exchange and the event "my edge planes of step n are computed"
void phase_begin(Shared &S, int g, int64_t n) {
   if (S.err.load()) return;
   const int k = (int)(n & 1);
   int rc = pf_engine_step_begin(S.eng[g], n);
   if (rc == PF_OK) rc = pf_engine_halo_ptrs(S.eng[g], &S.send_lo[k][g], &S.send_hi[k][g], &S.recv_lo[k][g], &S.recv_hi[k][g], nullptr);
   if (rc != PF_OK) S.set_error(rc, pf_last_error());
   else if (hipEventRecord(S.ev[k][g], S.edge[g]) != hipSuccess) S.set_error(PF_ERR_HIP, "hipEventRecord failed");
}
// B, peer copies (after every slab has done A): pull the neighbours' freshly computed edge planes into my ghost planes, on MY
// edge stream: ordered after my own edge kernels of this step (which were the last readers of the grid those ghost planes
// belong to) and after the neighbour's edge event; the interior keeps running on the main stream meanwhile
void phase_pull(Shared &S, int g, int64_t n) {
   const int k = (int)(n & 1), d = S.dev[g];
   auto pull = [&](int nb, void *dst, const void *src) {
      if (hipStreamWaitEvent(S.edge[g], S.ev[k][nb], 0) != hipSuccess) { S.set_error(PF_ERR_HIP, "hipStreamWaitEvent failed"); return; }
      const hipError_t e = (S.dev[nb] == d) ? hipMemcpyAsync(dst, src, S.plane_bytes, hipMemcpyDeviceToDevice, S.edge[g])
                                            : hipMemcpyPeerAsync(dst, d, src, S.dev[nb], S.plane_bytes, S.edge[g]);
      if (e != hipSuccess) S.set_error(PF_ERR_HIP, hipGetErrorString(e));
   };
   if (S.drop_step >= 0 && g == 1 && n == S.drop_step) return; // (test hook: the self-check must notice)
   if (S.only >= 0) { // (cost model of one rank: its own edge planes stand in for the neighbours')
      if (g > 0) pull(g, S.recv_lo[k][g], S.send_hi[k][g]);
      if (g < S.G - 1) pull(g, S.recv_hi[k][g], S.send_lo[k][g]);
      return;
   }
   if (g > 0) pull(g - 1, S.recv_lo[k][g], S.send_hi[k][g - 1]);         // left neighbour's last owned plane -> my plane 0
   if (g < S.G - 1) pull(g + 1, S.recv_hi[k][g], S.send_lo[k][g + 1]);   // right neighbour's first owned plane -> my last plane
}
// B, RCCL: slab g SENDS its two edge planes and RECEIVES its two ghost planes, all four operations in one group on its edge
// stream (ordered after its edge kernels: the sources are complete, and the last readers of the ghost planes are done).
// grouped: the caller has opened an ncclGroupStart that spans several slabs (one-thread mode: one host thread must not block
// in the group end of one slab before the matching operations of its neighbour are issued).
void phase_rccl(Shared &S, int g, int64_t n, bool grouped) {
   const int k = (int)(n & 1);
   const size_t nb = S.plane_bytes;
   if (S.rccl_self) {
      // 1-rank communicator: the neighbour's plane goes through RCCL to myself (after the neighbour's edge event)
      const int ql = S.only >= 0 ? g : g - 1, qh = S.only >= 0 ? g : g + 1; // (cost model of one rank: its own planes)
      for (int q : {ql, qh})
         if (q >= 0 && q < S.G && hipStreamWaitEvent(S.edge[g], S.ev[k][q], 0) != hipSuccess) { S.set_error(PF_ERR_HIP, "hipStreamWaitEvent failed"); return; }
      if (!grouped) NCHK(g, g_rccl.GroupStart());
      if (g > 0) { NCHK(g, g_rccl.Send(S.send_hi[k][ql], nb, ncclInt8, 0, S.comm[g], S.edge[g])); NCHK(g, g_rccl.Recv(S.recv_lo[k][g], nb, ncclInt8, 0, S.comm[g], S.edge[g])); }
      if (g < S.G - 1) { NCHK(g, g_rccl.Send(S.send_lo[k][qh], nb, ncclInt8, 0, S.comm[g], S.edge[g])); NCHK(g, g_rccl.Recv(S.recv_hi[k][g], nb, ncclInt8, 0, S.comm[g], S.edge[g])); }
      if (!grouped) NCHK(g, g_rccl.GroupEnd());
      return;
   }
   if (!grouped) NCHK(g, g_rccl.GroupStart());
   if (g > 0) {
      NCHK(g, g_rccl.Send(S.send_lo[k][g], nb, ncclInt8, S.rank[g - 1], S.comm[g], S.edge[g]));
      NCHK(g, g_rccl.Recv(S.recv_lo[k][g], nb, ncclInt8, S.rank[g - 1], S.comm[g], S.edge[g]));
   }
   if (g < S.G - 1) {
      NCHK(g, g_rccl.Send(S.send_hi[k][g], nb, ncclInt8, S.rank[g + 1], S.comm[g], S.edge[g]));
      NCHK(g, g_rccl.Recv(S.recv_hi[k][g], nb, ncclInt8, S.rank[g + 1], S.comm[g], S.edge[g]));
   }
   if (!grouped) NCHK(g, g_rccl.GroupEnd());
}
// host-staged, part 1 (sender, right after its edge event): my two edge planes -> my pinned buffer, on my edge stream.  The
// buffer of this parity was last read two steps ago: its readers' copies must be complete (host wait, normally long over).
void phase_stage(Shared &S, int g, int64_t n) {
   if (S.err.load()) return;
   const int k = (int)(n & 1);
   if (S.steps_done[g] >= 2)
      for (int nb : {g - 1, g + 1}) {
         if (S.only >= 0) nb = g;
         if (nb >= 0 && nb < S.G) MCHK(g, hipEventSynchronize(S.ev_h2d[k][nb]));
      }
   uint8_t *hb = (uint8_t *)S.hstage[k][g];
   MCHK(g, hipMemcpyAsync(hb, S.send_lo[k][g], S.plane_bytes, hipMemcpyDeviceToHost, S.edge[g]));
   MCHK(g, hipMemcpyAsync(hb + S.plane_bytes, S.send_hi[k][g], S.plane_bytes, hipMemcpyDeviceToHost, S.edge[g]));
   MCHK(g, hipEventRecord(S.ev_d2h[k][g], S.edge[g]));
}
// part 2 (receiver, after the barrier): wait ON THE HOST until the neighbour's planes are in its buffer, then copy them into my
// ghost planes on my edge stream (ordered after my own edge kernels, the last readers of those ghost planes)
void phase_unstage(Shared &S, int g, int64_t n) {
   const int k = (int)(n & 1);
   if (S.drop_step >= 0 && g == 1 && n == S.drop_step) { MCHK(g, hipEventRecord(S.ev_h2d[k][g], S.edge[g])); return; } // (test hook: the self-check must notice)
   const int ql = S.only >= 0 ? g : g - 1, qh = S.only >= 0 ? g : g + 1; // (cost model of one rank: its own planes)
   if (g > 0) {
      MCHK(g, hipEventSynchronize(S.ev_d2h[k][ql]));
      MCHK(g, hipMemcpyAsync(S.recv_lo[k][g], (const uint8_t *)S.hstage[k][ql] + S.plane_bytes, S.plane_bytes, hipMemcpyHostToDevice, S.edge[g])); // left neighbour's LAST owned plane
   }
   if (g < S.G - 1) {
      MCHK(g, hipEventSynchronize(S.ev_d2h[k][qh]));
      MCHK(g, hipMemcpyAsync(S.recv_hi[k][g], (const uint8_t *)S.hstage[k][qh], S.plane_bytes, hipMemcpyHostToDevice, S.edge[g]));                 // right neighbour's FIRST owned plane
   }
   MCHK(g, hipEventRecord(S.ev_h2d[k][g], S.edge[g]));
}
// B': exchange self-check, part 1: bit-pattern checksums of the two planes I sent and the two I received (after the edge
// stream has drained: the exchange of this step is complete on my side)
void phase_checksum(Shared &S, int g, int64_t n) {
   const int k = (int)(n & 1);
   MCHK(g, hipSetDevice(S.dev[g]));
   uint8_t *hb = S.hbuf[g].data();
   const void *src[4] = {S.send_lo[k][g], S.send_hi[k][g], S.recv_lo[k][g], S.recv_hi[k][g]};
   for (int i = 0; i < 4; i++) MCHK(g, hipMemcpyAsync(hb + (size_t)i * S.plane_bytes, src[i], S.plane_bytes, hipMemcpyDeviceToHost, S.edge[g]));
   MCHK(g, hipStreamSynchronize(S.edge[g]));
   for (int i = 0; i < 4; i++) {
      const uint32_t *w = (const uint32_t *)(hb + (size_t)i * S.plane_bytes);
      uint64_t sum = 0;
      for (size_t j = 0; j < S.plane_bytes / 4; j++) sum += (uint64_t)w[j] * (uint64_t)(1 + (j & 1023)); // position-weighted: a shifted plane does not pass
      S.sums[(size_t)g * 4 + i] = sum;
   }
}
// part 2 (after a barrier): what arrived in my ghost planes must be what my neighbours sent
void phase_compare(Shared &S, int g) {
   bool ok = true, nz = false;
   if (g > 0) { ok &= S.sums[(size_t)g * 4 + 2] == S.sums[(size_t)(g - 1) * 4 + 1]; nz |= S.sums[(size_t)g * 4 + 2] != 0; }
   if (g < S.G - 1) { ok &= S.sums[(size_t)g * 4 + 3] == S.sums[(size_t)(g + 1) * 4 + 0]; nz |= S.sums[(size_t)g * 4 + 3] != 0; }
   if (!ok) S.verify_bad.store(1);
   if (nz) S.verify_nonzero.store(1);
   if (g == 0) S.verify_checked.fetch_add(1);
}
// C: join the two streams, rotate the state
void phase_end(Shared &S, int g, int64_t n) {
   const int rc = pf_engine_step_end(S.eng[g], n);
   if (rc != PF_OK) S.set_error(rc, pf_last_error());
   S.steps_done[g]++;
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
   for (int k = 0; k < 2; k++) if (S.ev[k][g]) { hipEventDestroy(S.ev[k][g]); S.ev[k][g] = nullptr; }
   for (int k = 0; k < 5; k++) if (S.grids[k][g]) { hipFree(S.grids[k][g]); S.grids[k][g] = nullptr; }
   for (int k = 0; k < 2; k++) {
      if (!S.hstage[k].empty() && S.hstage[k][g]) { hipHostFree(S.hstage[k][g]); S.hstage[k][g] = nullptr; }
      if (!S.ev_d2h[k].empty() && S.ev_d2h[k][g]) { hipEventDestroy(S.ev_d2h[k][g]); S.ev_d2h[k][g] = nullptr; }
      if (!S.ev_h2d[k].empty() && S.ev_h2d[k][g]) { hipEventDestroy(S.ev_h2d[k][g]); S.ev_h2d[k][g] = nullptr; }
   }
}

// steps [n0, n0+ns) of slab g (its own host thread); returns when its streams have drained and its receivers are flushed
void run_slab(Shared &S, int g, int &local, int64_t n0, int64_t ns) {
   bool stop = S.err.load() != 0;
   for (int64_t n = n0; n < n0 + ns && !stop; n++) {
      hipSetDevice(S.dev[g]);
      const bool verify = S.steps_done[g] < S.verify_n; // the same decision in every thread: all slabs have done the same steps
      phase_begin(S, g, n);
      if (S.transport == TR_HOST) phase_stage(S, g, n);
      if ((S.faults & 4) && g == 1 && S.steps_done[g] == 3) // (test hook: this slab's thread hangs; the watchdog of the others must turn that into an error)
         std::this_thread::sleep_for(std::chrono::duration<double>(3.0 * S.bar_timeout + 1.0));
      stop = S.bar.wait(local, S.err, S.bar_timeout); // every slab's edge event of step n is recorded, its plane pointers published
      if (stop) break;
      if (S.transport == TR_RCCL) phase_rccl(S, g, n, false);
      else if (S.transport == TR_HOST) phase_unstage(S, g, n);
      else phase_pull(S, g, n);
      if (verify) {
         if (!S.err.load()) phase_checksum(S, g, n);
         stop = S.bar.wait(local, S.err, S.bar_timeout);
         if (stop) break;
         phase_compare(S, g);
      }
      phase_end(S, g, n);
   }
   if (S.bar.timed_out.load()) { // (the watchdog fired: report once, do not touch the devices any more -- a stuck stream would block the flush too)
      char b[200];
      snprintf(b, sizeof b, "slab chain hung: a slab's host thread did not reach the step barrier within %.0f s (a device or a collective is stuck; "
                            "PFFDTD_BARRIER_TIMEOUT_S sets the limit)", S.bar_timeout);
      S.set_error(PF_ERR_HIP, b);
      return;
   }
   finish_slab(S, g);
}

} // namespace

// the chain as an object: one persistent host thread per slab, commands posted by the caller's thread
struct pf_multi {
   Shared S;
   pf_simdata *sd = nullptr;
   std::vector<std::thread> th;
   std::mutex mu;
   std::condition_variable cv_cmd, cv_done;
   int64_t cmd_seq = 0, cmd_n0 = 0, cmd_ns = 0;
   int cmd_kind = 0;                 // 1 run, 2 quit
   int done = 0;
   bool created = false;
   double last_seconds = 0;
   bool one_thread = false;
   bool broken = false;              // the watchdog fired and a slab thread never came back: threads are abandoned, the object leaks
};

namespace {

void worker(pf_multi *m, int g) {
   Shared &S = m->S;
   int local = 0;
   create_slab(S, g);
   S.bar.wait(local, S.err, 10.0 * S.bar_timeout); // all engines exist (or an error is up)
   { std::lock_guard<std::mutex> lk(m->mu); m->done++; }
   m->cv_done.notify_all();
   int64_t seen = 0;
   for (;;) {
      int kind;
      int64_t n0, ns;
      {
         std::unique_lock<std::mutex> lk(m->mu);
         m->cv_cmd.wait(lk, [&] { return m->cmd_seq != seen; });
         seen = m->cmd_seq; kind = m->cmd_kind; n0 = m->cmd_n0; ns = m->cmd_ns;
      }
      if (kind == 2) break;
      run_slab(S, g, local, n0, ns);
      S.bar.wait(local, S.err, S.bar_timeout);
      { std::lock_guard<std::mutex> lk(m->mu); m->done++; }
      m->cv_done.notify_all();
   }
}

void post(pf_multi *m, int kind, int64_t n0, int64_t ns) {
   {
      std::lock_guard<std::mutex> lk(m->mu);
      m->cmd_kind = kind; m->cmd_n0 = n0; m->cmd_ns = ns; m->done = 0; m->cmd_seq++;
   }
   m->cv_cmd.notify_all();
}
void wait_done(pf_multi *m) {
   std::unique_lock<std::mutex> lk(m->mu);
   const int want = m->S.only >= 0 ? 1 : m->S.G;
   for (;;) {
      if (m->cv_done.wait_for(lk, std::chrono::milliseconds(200), [&] { return m->done == want; })) return;
      // the watchdog fired in the slab threads: give the healthy ones a moment to report, then stop waiting for the stuck one
      if (m->S.bar.timed_out.load()) {
         if (m->cv_done.wait_for(lk, std::chrono::duration<double>(std::min(5.0, m->S.bar_timeout)), [&] { return m->done == want; })) return;
         m->broken = true;
         return;
      }
   }
}

void scatter_outputs(pf_multi *m) { // receivers: every slab filled its own rows (gpu_engine.h:1066-1075)
   Shared &S = m->S;
   pf_simdata *sd = m->sd;
   if (!sd->u_out) return;
   for (int g = 0; g < S.G; g++) {
      const Slab &sl = S.slabs[g];
      for (size_t r = 0; r < sl.out_rows.size(); r++)
         memcpy(sd->u_out + sl.out_rows[r] * sd->Nt, sl.u_out.data() + r * (size_t)sd->Nt, sizeof(double) * (size_t)sd->Nt);
   }
}

const char *transport_name(const Shared &S) {
   if (S.G == 1) return "none (one slab)";
   if (S.transport == TR_HOST) return "host-staged";
   return S.transport == TR_RCCL ? (S.rccl_self ? "rccl (self-communicators, one device)" : "rccl") : "peer copies";
}

} // namespace

namespace {
// calibration chains (measure_wall_scale) are cut where the measurement wants them, not by partition()
thread_local const std::vector<int64_t> *tl_force_cuts = nullptr;
}

extern "C" {

int pf_slab_partition(const pf_simdata *sd, int32_t nslabs, int32_t even_split, int64_t *cuts) {
   return pf_slab_partition_w(sd, nslabs, even_split, 1.0, cuts);
}
int pf_slab_partition_w(const pf_simdata *sd, int32_t nslabs, int32_t even_split, double wall_scale, int64_t *cuts) {
   return pf_slab_partition_axis(sd, nslabs, even_split, wall_scale, 0, cuts);
}
int pf_slab_partition_axis(const pf_simdata *sd, int32_t nslabs, int32_t even_split, double wall_scale, int32_t along_z, int64_t *cuts) {
   if (!sd || !cuts) return fail("pf_slab_partition: null argument");
   if (!(wall_scale > 0)) wall_scale = 1.0;
   std::vector<int64_t> c;
   const int rc = partition(sd, nslabs, even_split != 0, c, along_z != 0, wall_scale);
   if (rc) return rc;
   for (int g = 0; g <= nslabs; g++) cuts[g] = c[g];
   return PF_OK;
}

// How much does a wall plane at the end of THIS scene's chain cost, in interior planes, on THIS device and build?  Three one-rank
// cost models (pf_opts.only_slab: a rank alone, its edge planes standing in for the neighbours'): an interior rank with p0 = Nx / G
// planes, one with p0 + dp planes (-> the cost of an interior plane), and the first rank with p0 planes (-> what its x wall adds).
// The ratio to what the compiled-in weights (23 / 5 interior planes per full plane of lossy / rigid nodes) predict for that wall is
// the factor partition() scales them by.  <= 0: not measured (scene too small, too few steps, a chain cut along file z, or a
// calibration run failed): the caller keeps factor 1.  Costs three short-lived slab engines (a few seconds at 1024^3 / 8).
static int multi_create_x(pf_simdata *sd, int32_t nslabs, const int32_t *devices, const pf_opts_x *base, pf_multi **out);
static double slab_wall_scale_x(pf_simdata *sd, int32_t nslabs, int32_t device, const pf_opts_x *base) {
   if (!sd || nslabs < 2 || tl_force_cuts) return -1.0;
   const int64_t Nx = sd->Nx, p0 = Nx / nslabs;
   const int64_t ncal = 63; // steps each calibration chain takes: 9 to warm up, then 18 timed, three times (the fastest counts)
   if (p0 < 24 || (sd->Npts / nslabs) < ((int64_t)1 << 24) || sd->Nt < ncal || 2 * p0 + 8 > Nx) return -1.0;
   std::vector<int64_t> c0;
   std::vector<double> wall1;
   if (partition(sd, nslabs, false, c0, false, 1.0, &wall1) != PF_OK) return -1.0;
   const int64_t dp = std::max<int64_t>(16, p0 / 2), cmid = std::max<int64_t>((Nx - p0 - dp) / 2, p0 + 1);
   if (cmid + p0 + dp + 2 > Nx) return -1.0;
   pf_opts_x o = *base;
   o.verify_exchange = 0; o.test_drop_exchange = 0; o.test_faults = 0; o.timing = 0;
   o.transport = PF_TRANSPORT_PEER;
   // The forced cuts are x planes.  The real chain is cut along file z when the caller forces that or when pf_multi_create would
   // choose it for G = nslabs (the same rule as there -- evaluated HERE, with the caller's G: a calibration chain of 2 or 3 slabs
   // could decide otherwise): no x wall to weigh then.  The calibration chains themselves are pinned to x.
   {
      const int vb = o.air_variant & 255;
      const bool can = !o.energy && vb != 40 && vb != 41 && !(o.multi_flags & PF_MULTI_FORCE_PAIRS) && nslabs < sd->Nz;
      if (o.multi_flags & PF_MULTI_CUT_Z) return -1.0;
      if (!(o.multi_flags & PF_MULTI_CUT_X) && !(o.debug & 0x2000) && o.layout != PF_LAYOUT_FILE && can && (sd->Nz - 2) / nslabs >= 16 && pf__axis_exchange_pays(sd, nullptr) != 0) return -1.0;
   }
   o.multi_flags = (o.multi_flags | PF_MULTI_CUT_X) & ~(PF_MULTI_CUT_Z | PF_MULTI_MEASURE_WEIGHTS | PF_MULTI_EVEN_SPLIT);
   auto one = [&](const std::vector<int64_t> &cuts, int slab) -> double {
      const int G = (int)cuts.size() - 1;
      std::vector<int32_t> devs(G, device);
      pf_opts_x oo = o;
      oo.only_slab = slab + 1;
      pf_multi *m = nullptr;
      tl_force_cuts = &cuts;
      const int rc = multi_create_x(sd, G, devs.data(), &oo, &m);
      tl_force_cuts = nullptr;
      if (rc != PF_OK || !m) return -1.0;
      double t = -1.0;
      if (pf_multi_run(m, 0, 9) == PF_OK) {
         for (int64_t n0 = 9; n0 + 18 <= ncal; n0 += 18) {
            pf_multi_info info;
            if (pf_multi_run(m, n0, 18) != PF_OK || pf_multi_get_info(m, &info) != PF_OK) { t = -1.0; break; }
            t = t < 0 ? info.last_run_seconds / 18.0 : std::min(t, info.last_run_seconds / 18.0);
         }
      }
      pf_multi_destroy(m);
      return t;
   };
   // the calibration chains write their (meaningless) receiver rows into sd->u_out like any chain: keep what was there
   std::vector<double> keep;
   if (sd->u_out && sd->Nr > 0) keep.assign(sd->u_out, sd->u_out + (size_t)sd->Nr * sd->Nt);
   struct Restore {
      pf_simdata *sd; std::vector<double> &keep;
      ~Restore() { if (!keep.empty()) memcpy(sd->u_out, keep.data(), sizeof(double) * keep.size()); }
   } restore{sd, keep};
   const double t1 = one({0, cmid, cmid + p0, Nx}, 1);
   const double t2 = t1 > 0 ? one({0, cmid, cmid + p0 + dp, Nx}, 1) : -1.0;
   const double te = t2 > 0 ? one({0, p0, Nx}, 0) : -1.0;
   if (!(t1 > 0 && t2 > 0 && te > 0)) return -1.0;
   double a = (t2 - t1) / (double)dp;                 // seconds per interior plane
   if (!(a > 0.25 * t1 / (double)p0)) a = t1 / (double)p0; // (noise: at least a quarter of the average cost per plane; else the average itself)
   const double w_planes = (te - t1) / a;             // what the first rank's wall adds, in interior planes
   double model = 0;
   for (int64_t x = 0; x < p0; x++) model += wall1[x] - wall1[cmid + x];
   if (!(model > 0.5)) return -1.0;                   // (no wall to speak of at the end of the chain)
   const double k = std::min(4.0, std::max(0.25, w_planes / model));
   if (getenv("PFFDTD_VERBOSE"))
      fprintf(stderr, "pffdtd_hip: wall weights measured on device %d: interior rank %ld planes %.4f ms, %ld planes %.4f ms, first rank %ld planes %.4f ms per step -> an interior "
                      "plane %.5f ms, the x wall %.1f interior planes (compiled-in weights: %.1f) -> factor %.2f\n", device, (long)p0, t1 * 1e3, (long)(p0 + dp), t2 * 1e3,
              (long)p0, te * 1e3, a * 1e3, w_planes, model, k);
   return k;
}

double pf_slab_wall_scale(pf_simdata *sd, int32_t nslabs, int32_t device, const pf_opts *base) {
   const pf_opts_x x = pf__take_hooks(base);
   return slab_wall_scale_x(sd, nslabs, device, &x);
}

int pf_multi_create(pf_simdata *sd, int32_t nslabs, const int32_t *devices, const pf_opts *base, pf_multi **out) {
   const pf_opts_x x = pf__take_hooks(base);
   return multi_create_x(sd, nslabs, devices, &x, out);
}
static int multi_create_x(pf_simdata *sd, int32_t nslabs, const int32_t *devices, const pf_opts_x *base, pf_multi **out) {
   if (!sd || nslabs < 1 || !devices || !out || !base) return fail("pf_multi_create: bad argument");
   *out = nullptr;
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { pf__set_error("no HIP device visible"); return PF_ERR_NODEV; }
   for (int g = 0; g < nslabs; g++)
      if (devices[g] < 0 || devices[g] >= ndev) return fail("pf_multi_create: device id out of range");
   pf_multi *m = new pf_multi();
   Shared &S = m->S;
   m->sd = sd;
   S.base = *base;
   const int G = nslabs;
   S.G = G; S.Nt = sd->Nt;
   S.dev.assign(devices, devices + G);
   S.eng.assign(G, nullptr);
   S.edge.assign(G, nullptr);
   S.paired.assign(G, 0);
   S.steps_done.assign(G, 0);
   for (int k = 0; k < 2; k++) {
      S.ev[k].assign(G, nullptr);
      S.send_lo[k].assign(G, nullptr); S.send_hi[k].assign(G, nullptr);
      S.recv_lo[k].assign(G, nullptr); S.recv_hi[k].assign(G, nullptr);
   }
   for (int k = 0; k < 5; k++) S.grids[k].assign(G, nullptr);
   for (int k = 0; k < 2; k++) { S.hstage[k].assign(G, nullptr); S.ev_d2h[k].assign(G, nullptr); S.ev_h2d[k].assign(G, nullptr); }
   S.bar_timeout = barrier_timeout();
   S.faults = S.base.test_faults;
   if (G == 1) { // plain single-domain engine
      pf_opts_x o = S.base;
      o.device = devices[0]; o.slab_first = o.slab_last = 1;
      S.cuts = {0, sd->Nx};
      const int rc = pf__engine_create_x(sd, &o, &S.eng[0]);
      if (rc != PF_OK) { delete m; return rc; }
      m->created = true;
      *out = m;
      return PF_OK;
   }
   // Which axis?  The reference cuts along x (gpu_engine.h:516-662).  Rooms whose engines would rather store the grid with the
   // x and z axes exchanged (+11-15 % on the reference's rooms) are cut along FILE Z instead: the slab axis
   // is then the exchanged storage's plane axis and ghost planes stay contiguous.  PF_MULTI_CUT_Z forces, PF_MULTI_CUT_X forbids.
   {
      const int vb = S.base.air_variant & 255;
      const bool can = !S.base.energy && vb != 40 && vb != 41 && !(S.base.multi_flags & PF_MULTI_FORCE_PAIRS) && G < sd->Nz;
      if (S.base.multi_flags & PF_MULTI_CUT_Z) {
         if (!can) { delete m; return fail("PF_MULTI_CUT_Z: single steps only, no energy diagnostic, fewer slabs than Nz"); }
         S.along_z = true;
      } else if (!(S.base.multi_flags & PF_MULTI_CUT_X) && !(S.base.debug & 0x2000) && S.base.layout != PF_LAYOUT_FILE && can && (sd->Nz - 2) / G >= 16)
         S.along_z = pf__axis_exchange_pays(sd, nullptr) != 0;
   }
   int rc = PF_OK;
   if (tl_force_cuts) { // (a calibration chain of pf_slab_wall_scale)
      if ((int)tl_force_cuts->size() != G + 1 || S.along_z) { delete m; return fail("internal: forced cuts do not fit the chain (they are x planes)"); }
      S.cuts = *tl_force_cuts;
   } else {
      // the wall planes' weights: the compiled-in figures times the caller's factor (pf_opts.wall_scale > 0), or times what three cost models
      // of this scene on this device and build give (PF_MULTI_MEASURE_WEIGHTS)
      const bool even = (S.base.multi_flags & PF_MULTI_EVEN_SPLIT) != 0;
      S.wall_scale = 1.0;
      if (S.base.wall_scale > 0) S.wall_scale = S.base.wall_scale;
      else if (!even && !S.along_z && (S.base.multi_flags & PF_MULTI_MEASURE_WEIGHTS)) {
         const double k = slab_wall_scale_x(sd, G, devices[0], &S.base);
         if (k > 0) { S.wall_scale = k; S.wall_measured = true; }
      }
      rc = partition(sd, G, even, S.cuts, S.along_z, S.wall_scale);
   }
   S.slabs.resize(G);
   for (int g = 0; g < G && rc == PF_OK; g++) rc = cut_slab(sd, S.cuts, g, G, S.slabs[g], S.along_z);
   S.plane_bytes = S.along_z ? pf_grid_bytes(1, sd->Ny, sd->Nx, sd->real_bytes) : pf_grid_bytes(1, sd->Ny, sd->Nz, sd->real_bytes);
   S.verify_n = S.base.verify_exchange > 0 ? S.base.verify_exchange : 0;
   if (const char *ev = getenv("PFFDTD_VERIFY_EXCHANGE")) S.verify_n = std::max(atoi(ev), 0);
   if (S.base.only_slab > 0) {
      if (S.base.only_slab > G) { delete m; return fail("pf_opts.only_slab: no such slab"); }
      S.only = S.base.only_slab - 1;
      S.verify_n = 0; // (nothing to compare with)
      for (int g = 0; g < G; g++) if (S.dev[g] != S.dev[S.only]) { delete m; return fail("pf_opts.only_slab: name one device for every slab"); }
   }
   if (S.base.test_drop_exchange > 0) { // fault injection for the tests: never without the check that must catch it
      S.drop_step = S.base.test_drop_exchange - 1;
      S.verify_n = std::max<int64_t>(S.verify_n, S.drop_step + 2);
   }
   S.sums.assign((size_t)G * 4, 0);
   S.hbuf.resize(G);
   if (rc == PF_OK) rc = choose_transport(S, S.base.transport);
   if (rc != PF_OK) { delete m; return rc; }
   S.bar.n = S.only >= 0 ? 1 : G;
   m->one_thread = (S.base.multi_flags & PF_MULTI_ONE_THREAD) != 0 && S.only < 0;
   if (m->one_thread) {
      // the reference's arrangement (one host thread drives every GPU, gpu_engine.h:993-1145): kept for debugging
      for (int g = 0; g < G && !S.err.load(); g++) create_slab(S, g);
   } else {
      for (int g = 0; g < G; g++) if (S.only < 0 || g == S.only) m->th.emplace_back(worker, m, g);
      wait_done(m);
      if (m->broken && !S.err.load()) S.set_error(PF_ERR_HIP, "slab chain hung while its engines were created");
   }
   m->created = true;
   if (S.err.load()) { const std::string keep = S.err_msg; const int code = S.err.load(); pf_multi_destroy(m); pf__set_error(keep.c_str()); return code; }
   if (getenv("PFFDTD_VERBOSE")) {
      fprintf(stderr, "pffdtd_hip: %d slabs%s, ghost planes by %s%s%s:", G, S.along_z ? " cut along file z (engines store the x and z axes exchanged)" : "", transport_name(S), S.transport == TR_RCCL ? ", " : "", S.transport == TR_RCCL ? g_rccl.where.c_str() : "");
      for (int g = 0; g < G; g++) fprintf(stderr, " [dev %d: planes %ld-%ld%s]", S.dev[g], (long)S.cuts[g], (long)S.cuts[g + 1] - 1, S.paired[g] == 3 ? ", triples" : (S.paired[g] ? ", pairs" : ""));
      fprintf(stderr, "\n");
   }
   *out = m;
   return PF_OK;
}

int pf_multi_run(pf_multi *m, int64_t n0, int64_t nsteps) {
   if (!m) return fail("pf_multi_run: null object");
   Shared &S = m->S;
   if (n0 < 0 || nsteps < 0 || n0 + nsteps > m->sd->Nt) return fail("pf_multi_run: steps outside [0, Nt)");
   const auto t0 = std::chrono::steady_clock::now();
   if (S.G == 1) {
      int rc = pf_engine_run(S.eng[0], n0, nsteps);
      if (rc == PF_OK) rc = pf_engine_sync(S.eng[0]);
      m->last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      return rc;
   }
   if (S.err.load()) { pf__set_error(S.err_msg.c_str()); return S.err.load(); }
   if (m->one_thread) {
      const int G = S.G;
      for (int64_t n = n0; n < n0 + nsteps && !S.err.load(); n++) {
         const bool verify = S.steps_done[0] < S.verify_n;
         for (int g = 0; g < G; g++) { hipSetDevice(S.dev[g]); phase_begin(S, g, n); if (S.transport == TR_HOST) phase_stage(S, g, n); }
         if (S.err.load()) break;
         if (S.transport == TR_RCCL) {
            bool open = g_rccl.GroupStart() == ncclSuccess;
            for (int g = 0; g < G && !S.err.load(); g++) { hipSetDevice(S.dev[g]); phase_rccl(S, g, n, true); }
            if (!open || g_rccl.GroupEnd() != ncclSuccess) S.set_error(PF_ERR_HIP, "RCCL group call failed");
         } else if (S.transport == TR_HOST) {
            for (int g = 0; g < G && !S.err.load(); g++) { hipSetDevice(S.dev[g]); phase_unstage(S, g, n); }
         } else {
            for (int g = 0; g < G && !S.err.load(); g++) { hipSetDevice(S.dev[g]); phase_pull(S, g, n); }
         }
         if (verify) {
            for (int g = 0; g < G && !S.err.load(); g++) phase_checksum(S, g, n);
            for (int g = 0; g < G && !S.err.load(); g++) phase_compare(S, g);
         }
         for (int g = 0; g < G && !S.err.load(); g++) { hipSetDevice(S.dev[g]); phase_end(S, g, n); }
      }
      for (int g = 0; g < G; g++) finish_slab(S, g);
   } else {
      post(m, 1, n0, nsteps);
      wait_done(m);
      if (m->broken && !S.err.load()) S.set_error(PF_ERR_HIP, "slab chain hung: a slab's host thread never came back");
   }
   m->last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
   if (S.err.load()) { pf__set_error(S.err_msg.c_str()); return S.err.load(); }
   scatter_outputs(m);
   return PF_OK;
}

int pf_multi_get_info(pf_multi *m, pf_multi_info *info) {
   if (!m || !info) return fail("pf_multi_get_info: null argument");
   const Shared &S = m->S;
   memset(info, 0, sizeof *info);
   info->nslabs = S.G;
   info->transport = S.G == 1 ? 0 : S.transport;
   info->rccl_self = S.rccl_self ? 1 : 0;
   info->exchanges_checked = S.verify_checked.load();
   info->exchange_verified = S.verify_checked.load() > 0 ? (S.verify_bad.load() ? 0 : 1) : -1;
   info->exchange_nonzero = S.verify_nonzero.load();
   info->cut_along_z = S.along_z ? 1 : 0;
   info->last_run_seconds = m->last_seconds;
   info->plane_bytes = (int64_t)S.plane_bytes;
   snprintf(info->transport_name, sizeof info->transport_name, "%s", transport_name(S));
   snprintf(info->transport_note, sizeof info->transport_note, "%s", S.transport_note.c_str());
   info->wall_scale = S.wall_scale;
   info->wall_measured = S.wall_measured ? 1 : 0;
   return PF_OK;
}

int pf_multi_get_slab(pf_multi *m, int32_t g, int64_t *x0, int64_t *x1, int32_t *device, int32_t *paired, pf_engine **engine) {
   if (!m || g < 0 || g >= m->S.G) return fail("pf_multi_get_slab: slab index out of range");
   const Shared &S = m->S;
   if (x0) *x0 = S.cuts[g];
   if (x1) *x1 = S.cuts[g + 1];
   if (device) *device = S.dev[g];
   if (paired) *paired = S.paired[g];
   if (engine) *engine = S.eng[g];
   return PF_OK;
}

void pf_multi_destroy(pf_multi *m) {
   if (!m) return;
   Shared &S = m->S;
   if (m->broken) { // a slab thread is stuck inside a driver / RCCL call: it cannot be joined and may still use the object -- abandon both
      for (auto &t : m->th) t.detach();
      return;
   }
   if (!m->th.empty()) {
      post(m, 2, 0, 0);
      for (auto &t : m->th) t.join();
   }
   if (S.G == 1) { if (S.eng[0]) { pf_engine_destroy(S.eng[0]); S.eng[0] = nullptr; } }
   else for (int g = 0; g < S.G; g++) destroy_slab(S, g);
   for (auto &c : S.comm) if (c) { g_rccl.CommDestroy(c); c = nullptr; }
   delete m;
}

// run_sim on a chain of slabs, slab g on device devices[g] (ids may repeat).  base: engine options common to all slabs
// (numerics, air_variant, readout_chunk, multi_flags, transport, verify_exchange); NULL = defaults.
double pf_run_sim_devices(pf_simdata *sd, int32_t nslabs, const int32_t *devices, const pf_opts *base) {
   pf_multi *m = nullptr;
   if (pf_multi_create(sd, nslabs, devices, base, &m) != PF_OK) return -1.0;
   const int rc = pf_multi_run(m, 0, sd->Nt);
   const double el = m->last_seconds;
   std::string keep = pf_last_error();
   const bool bad = m->S.verify_bad.load() != 0;
   pf_multi_destroy(m);
   if (rc != PF_OK) { pf__set_error(keep.c_str()); return -1.0; }
   if (bad) { pf__set_error("slab exchange self-check failed: a ghost plane does not hold what the neighbour sent"); return -1.0; }
   return el;
}

// ---- one process per device (pffdtd_amd/dist.py under torch.distributed.run): the slab's two planes by NATIVE ncclSend / ncclRecv on the
// engine's edge stream -- what the chain object above does for its slabs, for a host whose ranks are processes.  torch.distributed's own
// p2p (batch_isend_irecv) runs on a stream of its own behind two cross-stream hops: measured on one MI355X (a rank of 8 at 1024^3
// exchanging with itself, tools/host_loop_profile.py) every exchange sat between 130 us and 77 us of nothing, 0.33 ms per step where
// this path takes 0.23-0.25.  The host hands the unique id from rank 0 to the others (any channel; dist.py: broadcast_object_list).
struct pf_rccl_comm { ncclComm_t c = nullptr; int nranks = 0, rank = 0; };

int pf_rccl_unique_id(void *id128) {
   if (!id128) return fail("pf_rccl_unique_id: null argument");
   std::string why;
   { std::lock_guard<std::mutex> lk(g_rccl_mu); if (!g_rccl.load(why)) return fail("%s", why.c_str()); }
   static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId: 128 bytes");
   ncclUniqueId id;
   const ncclResult_t r = g_rccl.GetUniqueId(&id);
   if (r != ncclSuccess) return fail("ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
   memcpy(id128, &id, sizeof id);
   return PF_OK;
}
int pf_rccl_comm_create(const void *id128, int32_t nranks, int32_t rank, int32_t device, pf_rccl_comm **out) {
   if (!id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail("pf_rccl_comm_create: bad argument");
   *out = nullptr;
   std::string why;
   { std::lock_guard<std::mutex> lk(g_rccl_mu); if (!g_rccl.load(why)) return fail("%s", why.c_str()); }
   // on a helper thread with a timeout, like the chain's communicators: a rendezvous that never completes must not hang the caller
   struct Job { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t r = ncclSuccess; ncclComm_t c = nullptr; ncclUniqueId id; int n, rk, dev; };
   auto job = std::make_shared<Job>();
   memcpy(&job->id, id128, sizeof job->id); job->n = nranks; job->rk = rank; job->dev = device;
   fflush(stdout);
   const int saved_out = dup(1); // (RCCL's banner goes to stdout: a host that prints machine-readable results there must not find it in between)
   if (saved_out >= 0) dup2(2, 1);
   std::thread([job] {
      ncclResult_t r = ncclSuccess;
      if (hipSetDevice(job->dev) != hipSuccess) r = ncclUnhandledCudaError;
      if (r == ncclSuccess) r = g_rccl.CommInitRank(&job->c, job->n, job->id, job->rk);
      { std::lock_guard<std::mutex> l2(job->mu); job->r = r; job->done = true; }
      job->cv.notify_all();
   }).detach();
   double tmo = 60.0;
   if (const char *ev = getenv("PFFDTD_RCCL_INIT_TIMEOUT_S")) { const double v = atof(ev); if (v > 0) tmo = v; }
   bool finished;
   { std::unique_lock<std::mutex> l2(job->mu); finished = job->cv.wait_for(l2, std::chrono::duration<double>(tmo), [&] { return job->done; }); }
   if (saved_out >= 0) { fflush(stdout); dup2(saved_out, 1); close(saved_out); }
   if (!finished) { char b[160]; snprintf(b, sizeof b, "ncclCommInitRank (rank %d of %d) did not return within %.0f s", rank, nranks, tmo); pf__set_error(b); return PF_ERR_HIP; }
   if (job->r != ncclSuccess) { char b[256]; snprintf(b, sizeof b, "ncclCommInitRank (rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(job->r)); pf__set_error(b); return PF_ERR_HIP; }
   pf_rccl_comm *c = new pf_rccl_comm();
   c->c = job->c; c->nranks = nranks; c->rank = rank;
   *out = c;
   return PF_OK;
}
// between pf_engine_step_begin and pf_engine_step_end: my first / last updated plane to rank peer_lo / peer_hi, theirs into my ghost planes
// (peer < 0: no neighbour on that side), one group on the engine's edge stream -- ordered after the edge planes, before the next step's
int pf_rccl_exchange(pf_rccl_comm *c, pf_engine *e, int32_t peer_lo, int32_t peer_hi) {
   if (!c || !c->c || !e) return fail("pf_rccl_exchange: null argument");
   if (peer_lo >= c->nranks || peer_hi >= c->nranks) return fail("pf_rccl_exchange: no such rank");
   void *slo = nullptr, *shi = nullptr, *rlo = nullptr, *rhi = nullptr;
   size_t nb = 0;
   int rc = pf_engine_halo_ptrs(e, &slo, &shi, &rlo, &rhi, &nb);
   if (rc) return rc;
   hipStream_t s = (hipStream_t)pf_engine_stream(e, 1);
   ncclResult_t r = g_rccl.GroupStart();
   if (r == ncclSuccess && peer_lo >= 0) { r = g_rccl.Send(slo, nb, ncclChar, peer_lo, c->c, s); if (r == ncclSuccess) r = g_rccl.Recv(rlo, nb, ncclChar, peer_lo, c->c, s); }
   if (r == ncclSuccess && peer_hi >= 0) { r = g_rccl.Send(shi, nb, ncclChar, peer_hi, c->c, s); if (r == ncclSuccess) r = g_rccl.Recv(rhi, nb, ncclChar, peer_hi, c->c, s); }
   const ncclResult_t r2 = g_rccl.GroupEnd();
   if (r == ncclSuccess) r = r2;
   if (r != ncclSuccess) { char b[256]; snprintf(b, sizeof b, "pf_rccl_exchange: %s", g_rccl.GetErrorString(r)); pf__set_error(b); return PF_ERR_HIP; }
   return PF_OK;
}
void pf_rccl_comm_destroy(pf_rccl_comm *c) {
   if (!c) return;
   if (c->c) g_rccl.CommDestroy(c->c);
   delete c;
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
      // every slab should be worth a device: at least 16 planes and ~17 M cells of its own (the reference only asks for
      // ngpus < Nx, gpu_engine.h:682; a 3e6-cell grid cut eight ways is slower than on one GPU)
      n = (int)std::max<int64_t>(1, std::min<int64_t>(n, std::min<int64_t>((sd->Nx - 2) / 16, sd->Npts >> 24)));
      // Rooms (scenes the chain cuts along file z) run as TWO slabs per device when only one device is in use: the halves' kernels
      // overlap -- one half's boundary pass runs beside the other's interior kernel -- which a single domain's dependent launches
      // cannot.  Measured on one MI355X, whole step: CTK church 313 against 288 Gvox/s as one domain (3, 4, 6 slabs: 280, 279,
      // 199), Musikverein 354 against 341-345 (3, 4 slabs: 330, 351).  PFFDTD_DEVICES=0 (a chain of one named device) is one domain.
      int per_dev = 1;
      if (n == 1 && sd->Nz >= 64 && pf__axis_exchange_pays(sd, nullptr)) per_dev = 2;
      for (int i = 0; i < n; i++)
         for (int k = 0; k < per_dev; k++) devs.push_back(i);
   }
   pf_opts o;
   pf_opts_default(&o);
   o.verify_exchange = devs.size() > 1 ? 2 : 0; // first contact: the first two exchanges are checksummed
   return pf_run_sim_devices(sd, (int32_t)devs.size(), devs.data(), &o);
}

} // extern "C"
