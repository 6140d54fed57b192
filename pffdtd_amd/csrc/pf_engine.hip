// pf_engine.hip -- host side of libpffdtd_hip.so: the C ABI of include/pffdtd_hip.h over the kernels of
// pf_kernels.h.  Replaces the reference's `double run_sim(struct SimData*)` (c_cuda/gpu_engine.h:665-1255,
// c_cuda/cpu_engine.h:52-360) for MI355X.  Not derived from gpu_engine.h: different memory layout (padded
// pitch, engine-built skip-mask), different kernels (2.5D register marching), device-resident source signals
// and receiver ring instead of per-step host traffic, and a split-phase step for overlapped slab exchange.
#include "pf_engine_class.inc"

namespace {
thread_local std::string g_err;
}
std::mutex pf__tune_mu[64]; // creation-time measurements of engines that share a device run one at a time (both precisions)

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


struct pf_engine {
   pfeng::EngineBase *impl;
};
pfeng::EngineBase *pf__new_engine_f64(const pf_simdata *sd, const pf_opts_x *o, int *rc); // pf_engine_f64.hip

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

// development / test switches (pf_debug.h): pending for the next create call of this thread
static thread_local int32_t t_hooks[3] = {0, 0, 0};
void pf_internal_hooks(int32_t debug, int32_t test_drop_exchange, int32_t test_faults) {
   t_hooks[0] = debug; t_hooks[1] = test_drop_exchange; t_hooks[2] = test_faults;
}

int pf_engine_create(const pf_simdata *sd, const pf_opts *opts, pf_engine **out) {
   const pf_opts_x o = pf__take_hooks(opts);
   return pf__engine_create_x(sd, &o, out);
}

} // extern "C"

pf_opts_x pf__take_hooks(const pf_opts *opts) {
   pf_opts_x x{};
   if (opts) static_cast<pf_opts &>(x) = *opts; else pf_opts_default(&x);
   x.debug = t_hooks[0]; x.test_drop_exchange = t_hooks[1]; x.test_faults = t_hooks[2];
   t_hooks[0] = t_hooks[1] = t_hooks[2] = 0;
   if (const char *e = getenv("PFFDTD_DEBUG")) x.debug |= (int32_t)strtol(e, nullptr, 0);
   return x;
}

int pf__engine_create_x(const pf_simdata *sd, const pf_opts_x *opts, pf_engine **out) {
   if (!sd || !out || !opts) return set_err(PF_ERR_ARG, "null argument");
   *out = nullptr;
   pf_opts_x o = *opts;
   if (o.layout == PF_LAYOUT_EXCHANGED) o.debug |= PF_DBG_SWZ_ON; // (the engine's own switches: the same two bits the tests force)
   else if (o.layout == PF_LAYOUT_FILE) o.debug |= PF_DBG_SWZ_OFF;
   else if (o.layout != PF_LAYOUT_AUTO) return set_err(PF_ERR_ARG, "pf_opts.layout must be PF_LAYOUT_AUTO, _EXCHANGED or _FILE");
   pfeng::EngineBase *impl = nullptr;
   int rc;
   if (sd->real_bytes == 4) {
      auto *e = new Engine<float>();
      rc = e->init(sd, &o);
      impl = e;
#ifndef PF_DEV_F32_ONLY // (development builds, PFFDTD_DEV_F32=1 in pffdtd_amd/build.py: the fp32 unit alone; never shipped)
   } else if (sd->real_bytes == 8) {
      impl = pf__new_engine_f64(sd, &o, &rc);
#endif
   } else {
      return set_err(PF_ERR_ARG, "real_bytes must be 4 or 8 (got %d)", sd->real_bytes);
   }
   if (rc) { std::string keep = g_err; delete impl; g_err = keep; return rc; }
   *out = new pf_engine{impl};
   return PF_OK;
}

extern "C" {

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
