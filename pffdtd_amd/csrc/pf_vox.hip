// pf_vox.hip -- MI355X voxelizer behind include/pffdtd_vox.h.
//
// What the reference does (python/voxelizer/vox_scene.py:99-391): for every grid point near a surface and every
// leg k to one of its NN neighbours, cast a ray from the opposite neighbour through the point and find whether a
// triangle cuts the leg; points with a cut leg become boundary nodes, points lying on a surface lose all legs.
// The reference organises this as Python processes over a voxel hierarchy (vox_grid_base.py:64-199).
//
// Here: (1) k_bin bins triangles into fixed cells of 4x4x16 grid points (bounding box + plane-slab test, both
// conservative), twice: count, then fill; (2) k_vox runs one workgroup per non-empty cell, one thread per grid
// point, candidate triangles staged through LDS, all NN rays of a (point, triangle) pair in registers.  Every
// comparison that decides the output uses the reference's expression with the same operand order in IEEE
// double (this file is compiled with -ffp-contract=off), so the outputs are identical, not just close.
// The result does not depend on the order candidates are visited in: cut bits and the on-surface flag are ORs,
// and the nearest triangle is the smallest (distance, triangle index) -- the reference visits triangles in
// ascending index and replaces on strictly smaller distance only (vox_scene.py:226-230), which is the same thing.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "pffdtd_hip.h"
#include "pffdtd_vox.h"

#pragma clang fp contract(off)

extern "C" void pf__set_error(const char *msg); // pf_engine.hip (feeds pf_last_error)

namespace pfv {

constexpr int CX = 4, CY = 4, CZ = 16, CP = CX * CY * CZ; // grid points per cell = threads per workgroup
constexpr int TB = 32;                                    // triangles staged per LDS batch
constexpr int TD = PF_VOX_TRI_DOUBLES;

struct Grid {
   int64_t Nx, Ny, Nz;
   int ncx, ncy, ncz;
   double x0, y0, z0, h;
   const double *xv, *yv, *zv;
};

struct Legs {
   double vvh[12][3];
   double run[12][3];
   double hf, hfe, hf1, nb_eps, d_eps, cp_eps;
   int fcc;
};

// inclusive range of grid-point indices in [1, N-2] whose coordinate may lie in [lo, hi] (two points of slack)
__device__ inline bool axis_range(double lo, double hi, double v0, double h, int64_t N, int C, int &c0, int &c1) {
   int64_t i0 = (int64_t)floor((lo - v0) / h) - 1, i1 = (int64_t)floor((hi - v0) / h) + 2;
   i0 = i0 < 1 ? 1 : i0;
   i1 = i1 > N - 2 ? N - 2 : i1;
   if (i1 < i0) return false;
   c0 = (int)((i0 - 1) / C);
   c1 = (int)((i1 - 1) / C);
   return true;
}

// FILL=false: count candidates per cell; FILL=true: write the triangle index into the cell's slot range
template <bool FILL>
__global__ void __launch_bounds__(256) k_bin(Grid g, const double *__restrict__ tris, int64_t ntris, double hfe,
                                            int32_t *__restrict__ cnt, const int64_t *__restrict__ off,
                                            int32_t *__restrict__ list) {
   const int64_t ti = blockIdx.x;
   if (ti >= ntris) return;
   const double *T = tris + ti * TD;
   int cx0, cx1, cy0, cy1, cz0, cz1;
   if (!axis_range(T[24], T[27], g.x0, g.h, g.Nx, CX, cx0, cx1)) return;
   if (!axis_range(T[25], T[28], g.y0, g.h, g.Ny, CY, cy0, cy1)) return;
   if (!axis_range(T[26], T[29], g.z0, g.h, g.Nz, CZ, cz0, cz1)) return;
   const int nx = cx1 - cx0 + 1, ny = cy1 - cy0 + 1, nz = cz1 - cz0 + 1;
   const int64_t total = (int64_t)nx * ny * nz;
   const double n0 = T[3], n1 = T[4], n2 = T[5];
   for (int64_t c = threadIdx.x; c < total; c += blockDim.x) {
      const int cz = cz0 + (int)(c % nz), cy = cy0 + (int)((c / nz) % ny), cx = cx0 + (int)(c / ((int64_t)nz * ny));
      const int64_t ia = 1 + (int64_t)cx * CX, ja = 1 + (int64_t)cy * CY, ka = 1 + (int64_t)cz * CZ;
      const int64_t ib = min(ia + CX - 1, g.Nx - 2), jb = min(ja + CY - 1, g.Ny - 2), kb = min(ka + CZ - 1, g.Nz - 2);
      const double xa = g.xv[ia], xb = g.xv[ib], ya = g.yv[ja], yb = g.yv[jb], za = g.zv[ka], zb = g.zv[kb];
      if (xb < T[24] || xa > T[27] || yb < T[25] || ya > T[28] || zb < T[26] || za > T[29]) continue;
      // slab around the triangle's plane: every point that can pass |dotv(unor, cent-xyz)| <= hfe lies in it
      const double ccx = 0.5 * (xa + xb), ccy = 0.5 * (ya + yb), ccz = 0.5 * (za + zb);
      const double dist = fabs(n0 * (ccx - T[0]) + n1 * (ccy - T[1]) + n2 * (ccz - T[2]));
      const double rad = fabs(n0) * 0.5 * (xb - xa) + fabs(n1) * 0.5 * (yb - ya) + fabs(n2) * 0.5 * (zb - za);
      if (dist > rad + hfe * 1.000001 + 1e-9 * g.h) continue;
      const int64_t cell = ((int64_t)cx * g.ncy + cy) * g.ncz + cz;
      const int32_t slot = atomicAdd(&cnt[cell], 1);
      if (FILL) list[off[cell] + slot] = (int32_t)ti;
   }
}

__device__ inline double dot3(double a0, double a1, double a2, double b0, double b1, double b2) {
   return (a0 * b0 + a1 * b1) + a2 * b2; // np.sum(a*b, axis=-1) over three elements
}

template <int NN>
__global__ void __launch_bounds__(CP) k_vox(Grid g, Legs L, const double *__restrict__ tris,
                                           const int32_t *__restrict__ cells, const int64_t *__restrict__ off,
                                           const int32_t *__restrict__ list, int64_t *__restrict__ o_idx,
                                           uint16_t *__restrict__ o_cut, int32_t *__restrict__ o_tidx,
                                           double *__restrict__ o_nd, unsigned long long *__restrict__ o_count) {
   __shared__ double sT[TB * TD];
   __shared__ int32_t sI[TB];
   const int cell = cells[blockIdx.x];
   const int cz = cell % g.ncz, cy = (cell / g.ncz) % g.ncy, cx = cell / (g.ncz * g.ncy);
   const int t = threadIdx.x;
   const int64_t ix = 1 + (int64_t)cx * CX + t / (CY * CZ), iy = 1 + (int64_t)cy * CY + (t / CZ) % CY,
                 iz = 1 + (int64_t)cz * CZ + t % CZ;
   bool live = ix <= g.Nx - 2 && iy <= g.Ny - 2 && iz <= g.Nz - 2;
   if (L.fcc && ((ix + iy + iz) & 1)) live = false;
   double x = 0, y = 0, z = 0;
   if (live) { x = g.xv[ix]; y = g.yv[iy]; z = g.zv[iz]; }
   const double INF = __builtin_huge_val();
   unsigned cut = 0;
   bool nb = false;
   double ndist = INF;
   int32_t tidx = -1;
   const int64_t b0 = off[cell], b1 = off[cell + 1];
   for (int64_t base = b0; base < b1; base += TB) {
      const int nbatch = (int)min<int64_t>(TB, b1 - base);
      __syncthreads();
      if (t < nbatch) sI[t] = list[base + t];
      __syncthreads();
      for (int i = t; i < nbatch * TD; i += CP) sT[i] = tris[(int64_t)sI[i / TD] * TD + i % TD];
      __syncthreads();
      if (!live) continue;
      for (int j = 0; j < nbatch; j++) {
         const double *T = sT + j * TD;
         // bounding box +- hfe (vox_scene.py:170-171)
         if (!(x >= T[24] && y >= T[25] && z >= T[26] && x <= T[27] && y <= T[28] && z <= T[29])) continue;
         // distance to the triangle's plane (vox_scene.py:177-179)
         const double dtp = dot3(T[3], T[4], T[5], T[0] - x, T[1] - y, T[2] - z);
         if (!(fabs(dtp) <= L.hfe)) continue;
         const int32_t ti = sI[j];
         bool tnb = false;
#pragma unroll
         for (int k = 0; k < NN; k++) {
            // tri_ray_intersection_vec (tri_ray_intersection.py:67-107), ray from the opposite neighbour
            const double ox = x - L.vvh[k][0], oy = y - L.vvh[k][1], oz = z - L.vvh[k][2];
            const double r0 = L.run[k][0], r1 = L.run[k][1], r2 = L.run[k][2];
            double beta = dot3(r0, r1, r2, T[3], T[4], T[5]);
            bool fail = fabs(beta) < L.cp_eps;
            if (fail) beta = -2.220446049250313e-16;
            const double tt = dot3(T[3], T[4], T[5], T[0] - ox, T[1] - oy, T[2] - oz) / beta;
            fail |= tt < 0;
            const double px = ox + r0 * tt, py = oy + r1 * tt, pz = oz + r2 * tt;
            fail |= dot3(px - T[6], py - T[7], pz - T[8], T[15], T[16], T[17]) > L.d_eps;
            fail |= dot3(px - T[9], py - T[10], pz - T[11], T[18], T[19], T[20]) > L.d_eps;
            fail |= dot3(px - T[12], py - T[13], pz - T[14], T[21], T[22], T[23]) > L.d_eps;
            double hd = fail ? INF : tt;
            hd -= L.hf;                                   // distance past the point itself (vox_scene.py:209-210)
            if (hd < -L.nb_eps) hd = INF;                 // behind the point
            if (fabs(hd) <= L.nb_eps) tnb = true;         // the point lies on this triangle
            if (tnb) hd = fabs(hd);
            if (hd > L.hf1) hd = INF;
            if (hd <= L.hf1) {                            // leg k is cut (vox_scene.py:222-232)
               cut |= 1u << k;
               if (hd < ndist || (hd == ndist && ti < tidx)) { ndist = hd; tidx = ti; }
            }
         }
         nb |= tnb;
      }
   }
   if (nb) cut = (1u << NN) - 1;                          // on-surface points lose every leg (vox_scene.py:238)
   if (live && cut) {
      const unsigned long long s = atomicAdd(o_count, 1ull);
      o_idx[s] = (ix * g.Ny + iy) * g.Nz + iz;
      o_cut[s] = (uint16_t)cut;
      o_tidx[s] = tidx;
      o_nd[s] = ndist;
   }
}

} // namespace pfv

struct pf_vox_job {
   std::vector<int64_t> idx;
   std::vector<uint16_t> cut;
   std::vector<int32_t> tidx;
   std::vector<double> nd;
   pf_vox_stats st{};
};

namespace {

int vfail(const char *fmt, const char *a = "", long b = 0) {
   char buf[512];
   snprintf(buf, sizeof buf, fmt, a, b);
   pf__set_error(buf);
   return PF_ERR_ARG;
}

struct DevBuf {
   void *p = nullptr;
   ~DevBuf() { if (p) hipFree(p); }
   template <typename T> T *as() { return (T *)p; }
   hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
};

#define VCHK(expr)                                                                                              \
   do {                                                                                                         \
      hipError_t _e = (expr);                                                                                   \
      if (_e != hipSuccess) {                                                                                   \
         char _b[512];                                                                                          \
         snprintf(_b, sizeof _b, "HIP error %s at %s:%d: %s", hipGetErrorName(_e), __FILE__, __LINE__,          \
                  hipGetErrorString(_e));                                                                       \
         pf__set_error(_b);                                                                                     \
         return PF_ERR_HIP;                                                                                     \
      }                                                                                                         \
   } while (0)

int run_impl(const pf_vox_desc &d, pf_vox_job &job) {
   using namespace pfv;
   if (d.Nx < 3 || d.Ny < 3 || d.Nz < 3) return vfail("grid must be at least 3x3x3%s", "");
   if (!((d.NN == 6 && !d.fcc) || (d.NN == 12 && d.fcc))) return vfail("NN must be 6 (Cartesian) or 12 (FCC)%s", "");
   if (!d.xv || !d.yv || !d.zv || !d.vvh || !d.ray_un || (d.Ntris > 0 && !d.tris)) return vfail("null pointer in pf_vox_desc%s", "");
   if (!(d.h > 0) || !(d.hf > 0)) return vfail("grid spacing must be positive%s", "");
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { pf__set_error("no HIP device visible"); return PF_ERR_NODEV; }
   if (d.device < 0 || d.device >= ndev) return vfail("device %s%ld out of range", "", (long)d.device);
   VCHK(hipSetDevice(d.device));
   const auto t_start = std::chrono::steady_clock::now();

   Grid g;
   g.Nx = d.Nx; g.Ny = d.Ny; g.Nz = d.Nz; g.h = d.h;
   g.x0 = d.xv[0]; g.y0 = d.yv[0]; g.z0 = d.zv[0];
   g.ncx = (int)((d.Nx - 2 + CX - 1) / CX); g.ncy = (int)((d.Ny - 2 + CY - 1) / CY); g.ncz = (int)((d.Nz - 2 + CZ - 1) / CZ);
   const int64_t ncells = (int64_t)g.ncx * g.ncy * g.ncz;
   if (ncells >= ((int64_t)1 << 31)) return vfail("grid too large for 32-bit cell ids%s", "");
   Legs L;
   memset(&L, 0, sizeof L);
   for (int k = 0; k < d.NN; k++)
      for (int a = 0; a < 3; a++) { L.vvh[k][a] = d.vvh[k * 3 + a]; L.run[k][a] = d.ray_un[k * 3 + a]; }
   L.hf = d.hf; L.hfe = d.hfe; L.hf1 = d.hf1; L.nb_eps = d.nb_eps; L.d_eps = d.d_eps; L.cp_eps = d.cp_eps; L.fcc = d.fcc;

   DevBuf bx, by, bz, btri, bcnt, boff, blist, bcells;
   VCHK(bx.alloc(sizeof(double) * d.Nx)); VCHK(by.alloc(sizeof(double) * d.Ny)); VCHK(bz.alloc(sizeof(double) * d.Nz));
   VCHK(hipMemcpy(bx.p, d.xv, sizeof(double) * d.Nx, hipMemcpyHostToDevice));
   VCHK(hipMemcpy(by.p, d.yv, sizeof(double) * d.Ny, hipMemcpyHostToDevice));
   VCHK(hipMemcpy(bz.p, d.zv, sizeof(double) * d.Nz, hipMemcpyHostToDevice));
   g.xv = bx.as<double>(); g.yv = by.as<double>(); g.zv = bz.as<double>();
   VCHK(btri.alloc(sizeof(double) * TD * d.Ntris));
   if (d.Ntris) VCHK(hipMemcpy(btri.p, d.tris, sizeof(double) * TD * d.Ntris, hipMemcpyHostToDevice));
   VCHK(bcnt.alloc(sizeof(int32_t) * ncells));
   VCHK(hipMemset(bcnt.p, 0, sizeof(int32_t) * ncells));

   hipEvent_t e0, e1, e2;
   VCHK(hipEventCreate(&e0)); VCHK(hipEventCreate(&e1)); VCHK(hipEventCreate(&e2));
   float ms_bin = 0, ms_vox = 0, ms = 0;
   std::vector<int32_t> cnt((size_t)ncells);
   std::vector<int64_t> off((size_t)ncells + 1, 0);
   std::vector<int32_t> nonempty;
   if (d.Ntris) {
      VCHK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_bin<false>, dim3((unsigned)d.Ntris), dim3(256), 0, 0, g, btri.as<double>(), d.Ntris, d.hfe,
                         bcnt.as<int32_t>(), (const int64_t *)nullptr, (int32_t *)nullptr);
      VCHK(hipEventRecord(e1, 0));
      VCHK(hipMemcpy(cnt.data(), bcnt.p, sizeof(int32_t) * ncells, hipMemcpyDeviceToHost));
      VCHK(hipEventElapsedTime(&ms, e0, e1));
      ms_bin += ms;
   }
   for (int64_t c = 0; c < ncells; c++) {
      off[c + 1] = off[c] + cnt[c];
      if (cnt[c]) nonempty.push_back((int32_t)c);
   }
   const int64_t npairs = off[ncells];
   job.st.ncells = ncells; job.st.ncells_nonempty = (int64_t)nonempty.size(); job.st.npairs = npairs;
   job.st.npoints_tested = (int64_t)nonempty.size() * CP;
   if (npairs > 0) {
      VCHK(boff.alloc(sizeof(int64_t) * (ncells + 1)));
      VCHK(hipMemcpy(boff.p, off.data(), sizeof(int64_t) * (ncells + 1), hipMemcpyHostToDevice));
      VCHK(blist.alloc(sizeof(int32_t) * npairs));
      VCHK(hipMemset(bcnt.p, 0, sizeof(int32_t) * ncells));
      VCHK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k_bin<true>, dim3((unsigned)d.Ntris), dim3(256), 0, 0, g, btri.as<double>(), d.Ntris, d.hfe,
                         bcnt.as<int32_t>(), boff.as<int64_t>(), blist.as<int32_t>());
      VCHK(hipEventRecord(e1, 0));
      VCHK(hipEventSynchronize(e1));
      VCHK(hipEventElapsedTime(&ms, e0, e1));
      ms_bin += ms;
      VCHK(bcells.alloc(sizeof(int32_t) * nonempty.size()));
      VCHK(hipMemcpy(bcells.p, nonempty.data(), sizeof(int32_t) * nonempty.size(), hipMemcpyHostToDevice));

      // ray-triangle kernel over the non-empty cells, in batches that bound the output buffers
      const int64_t BATCH = (int64_t)1 << 18;
      const int64_t cap = std::min<int64_t>(BATCH, (int64_t)nonempty.size()) * CP;
      DevBuf oi, oc, ot, on, ocount;
      VCHK(oi.alloc(sizeof(int64_t) * cap)); VCHK(oc.alloc(sizeof(uint16_t) * cap)); VCHK(ot.alloc(sizeof(int32_t) * cap));
      VCHK(on.alloc(sizeof(double) * cap)); VCHK(ocount.alloc(sizeof(unsigned long long)));
      for (int64_t c0 = 0; c0 < (int64_t)nonempty.size(); c0 += BATCH) {
         const int64_t nc = std::min<int64_t>(BATCH, (int64_t)nonempty.size() - c0);
         VCHK(hipMemset(ocount.p, 0, sizeof(unsigned long long)));
         VCHK(hipEventRecord(e1, 0));
         if (d.NN == 6)
            hipLaunchKernelGGL(k_vox<6>, dim3((unsigned)nc), dim3(CP), 0, 0, g, L, btri.as<double>(), bcells.as<int32_t>() + c0,
                               boff.as<int64_t>(), blist.as<int32_t>(), oi.as<int64_t>(), oc.as<uint16_t>(), ot.as<int32_t>(),
                               on.as<double>(), ocount.as<unsigned long long>());
         else
            hipLaunchKernelGGL(k_vox<12>, dim3((unsigned)nc), dim3(CP), 0, 0, g, L, btri.as<double>(), bcells.as<int32_t>() + c0,
                               boff.as<int64_t>(), blist.as<int32_t>(), oi.as<int64_t>(), oc.as<uint16_t>(), ot.as<int32_t>(),
                               on.as<double>(), ocount.as<unsigned long long>());
         VCHK(hipGetLastError());
         VCHK(hipEventRecord(e2, 0));
         unsigned long long n = 0;
         VCHK(hipMemcpy(&n, ocount.p, sizeof n, hipMemcpyDeviceToHost));
         VCHK(hipEventElapsedTime(&ms, e1, e2));
         ms_vox += ms;
         const size_t o = job.idx.size();
         job.idx.resize(o + n); job.cut.resize(o + n); job.tidx.resize(o + n); job.nd.resize(o + n);
         if (n) {
            VCHK(hipMemcpy(job.idx.data() + o, oi.p, sizeof(int64_t) * n, hipMemcpyDeviceToHost));
            VCHK(hipMemcpy(job.cut.data() + o, oc.p, sizeof(uint16_t) * n, hipMemcpyDeviceToHost));
            VCHK(hipMemcpy(job.tidx.data() + o, ot.p, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
            VCHK(hipMemcpy(job.nd.data() + o, on.p, sizeof(double) * n, hipMemcpyDeviceToHost));
         }
      }
   }
   hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2);
   job.st.ms_bin = ms_bin; job.st.ms_vox = ms_vox;
   job.st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
   return PF_OK;
}

} // namespace

extern "C" {

pf_vox_job *pf_vox_run(const pf_vox_desc *desc) {
   if (!desc) { pf__set_error("pf_vox_run: null descriptor"); return nullptr; }
   pf_vox_job *job = new pf_vox_job();
   if (run_impl(*desc, *job) != PF_OK) { delete job; return nullptr; }
   return job;
}

int64_t pf_vox_count(const pf_vox_job *job) { return job ? (int64_t)job->idx.size() : -1; }

int pf_vox_fetch(const pf_vox_job *job, int64_t *bn_ixyz, uint16_t *cut_bits, int32_t *tidx, double *ndist) {
   if (!job) { pf__set_error("pf_vox_fetch: null job"); return PF_ERR_ARG; }
   const size_t n = job->idx.size();
   if (bn_ixyz) memcpy(bn_ixyz, job->idx.data(), n * sizeof(int64_t));
   if (cut_bits) memcpy(cut_bits, job->cut.data(), n * sizeof(uint16_t));
   if (tidx) memcpy(tidx, job->tidx.data(), n * sizeof(int32_t));
   if (ndist) memcpy(ndist, job->nd.data(), n * sizeof(double));
   return PF_OK;
}

int pf_vox_get_stats(const pf_vox_job *job, pf_vox_stats *st) {
   if (!job || !st) { pf__set_error("pf_vox_get_stats: null argument"); return PF_ERR_ARG; }
   *st = job->st;
   return PF_OK;
}

void pf_vox_free(pf_vox_job *job) { delete job; }

} // extern "C"
