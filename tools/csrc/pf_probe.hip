// pf_probe.hip -- tools-only library (tools/libpf_probe.so): memory-system calibration and temporal-blocking research
// probes.  Not linked into, and not needed by, libpffdtd_hip.so.  Built by pffdtd_amd.build.build_probe().
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

#include "pf_probe.h"
#include "pf_probe_kernels.h"
#include "pf_tb3.h"

namespace {
std::string g_perr;
void probe_err(const char *msg) { g_perr = msg; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t grid_pitch(int64_t Nz, int32_t real_bytes) { return (Nz + 128 / real_bytes - 1) / (128 / real_bytes) * (128 / real_bytes); }
} // namespace

template <int R, int WY, int PF, int MODE = 0, int WZ = 1> static void membench_launch(float *u0, float *u1, pf::LeanParams fp, hipStream_t s) {
   fp.nyt = (int)cdiv(fp.Ny - 2, (int64_t)WY * R);
   fp.nzt = (int)cdiv(fp.P, 256 * WZ);
   dim3 g((uint32_t)fp.nzt * fp.nyt * fp.nxc), b(64 * WY * WZ);
   hipLaunchKernelGGL((pf::k_march_stream<float, R, WY, PF, MODE, WZ>), g, b, 0, s, u1, u0, fp);
}


extern "C" {

const char *pf_probe_last_error(void) { return g_perr.c_str(); }

double pf_membench(void *u0v, void *u1v, int64_t Nx, int64_t Ny, int64_t Nz, int32_t kind, int32_t R, int32_t WY,
                   int32_t PF, int32_t chunk, int32_t swizzle, int32_t reps) {
   float *u0 = (float *)u0v, *u1 = (float *)u1v;
   const int64_t P = grid_pitch(Nz, 4);
   pf::LeanParams fp{};
   fp.plane = Ny * P; fp.Nx = (int)Nx; fp.Ny = (int)Ny; fp.Nz = (int)Nz; fp.P = (int)P;
   fp.x_begin = 1; fp.x_end = (int)Nx - 1;
   fp.chunk = chunk > 0 ? chunk : 128;
   fp.nxc = (int)cdiv(Nx - 2, fp.chunk);
   fp.nzt = (int)cdiv(P, 256);
   fp.swizzle = swizzle;
   hipEvent_t e0, e1;
   hipEventCreate(&e0); hipEventCreate(&e1);
   auto launch = [&]() {
      if (kind == 0) {
         const int64_t nvec = Nx * Ny * P / 4;
         // R encodes MODE (bit0 nt loads, bit1 nt stores, bit2 one-shot), WY the unroll
#define PF_LS(m, u) if (R == m && WY == u) { const int64_t per = 256LL * u; const unsigned nb = (m & 4) ? (unsigned)cdiv(nvec, per) : 256u * 16u; hipLaunchKernelGGL((pf::k_linear_stream<float, m, u>), dim3(nb), dim3(256), 0, 0, u1, u0, nvec); return true; }
         PF_LS(0, 1) PF_LS(1, 1) PF_LS(2, 1) PF_LS(3, 1) PF_LS(4, 1) PF_LS(7, 1) PF_LS(0, 4) PF_LS(3, 4) PF_LS(4, 4) PF_LS(7, 4) PF_LS(6, 4) PF_LS(5, 4) PF_LS(4, 2) PF_LS(7 + 8 * 3, 4) PF_LS(7 + 8 * 6, 4) PF_LS(7 + 8 * 8, 4) PF_LS(7 + 8 * 10, 4) PF_LS(4 + 8 * 3, 4) PF_LS(4 + 8 * 8, 4)
#undef PF_LS
         return false;
      }
#define PF_MB(r, wy, pfd) if (kind == 1 && R == r && WY == wy && PF == pfd) { membench_launch<r, wy, pfd>(u0, u1, fp, 0); return true; }
      PF_MB(4, 4, 1) PF_MB(4, 4, 2) PF_MB(4, 4, 3) PF_MB(2, 4, 1) PF_MB(2, 4, 2) PF_MB(2, 4, 4) PF_MB(1, 4, 2) PF_MB(1, 4, 4) PF_MB(1, 4, 8)
      PF_MB(2, 8, 2) PF_MB(4, 8, 2) PF_MB(8, 4, 1) PF_MB(8, 4, 2) PF_MB(1, 8, 4)
#undef PF_MB
      // kind = 1 + MODE (bit0 nt u1 loads, bit1 nt u0 loads, bit2 nt stores) for R=4, WY=4, PF=1
#define PF_MM(m) if (kind == 1 + m && R == 4 && WY == 4 && PF == 1) { membench_launch<4, 4, 1, m>(u0, u1, fp, 0); return true; }
      PF_MM(1) PF_MM(2) PF_MM(3) PF_MM(4) PF_MM(5) PF_MM(6) PF_MM(7)
#undef PF_MM
      // kind 20 + m: full-row tiles (WZ=4 waves side by side), R rows, WY=1|2, nt mode m (0 or 7)
      if (kind == 20 && R == 4 && WY == 1) { membench_launch<4, 1, 1, 0, 4>(u0, u1, fp, 0); return true; }
      if (kind == 27 && R == 4 && WY == 1) { membench_launch<4, 1, 1, 7, 4>(u0, u1, fp, 0); return true; }
      if (kind == 20 && R == 4 && WY == 2) { membench_launch<4, 2, 1, 0, 4>(u0, u1, fp, 0); return true; }
      if (kind == 27 && R == 4 && WY == 2) { membench_launch<4, 2, 1, 7, 4>(u0, u1, fp, 0); return true; }
      if (kind == 20 && R == 2 && WY == 2) { membench_launch<2, 2, 1, 0, 4>(u0, u1, fp, 0); return true; }
      if (kind == 27 && R == 2 && WY == 2) { membench_launch<2, 2, 1, 7, 4>(u0, u1, fp, 0); return true; }
      if (kind == 27 && R == 1 && WY == 4) { membench_launch<1, 4, 2, 7, 4>(u0, u1, fp, 0); return true; }
      if (kind == 27 && R == 8 && WY == 1) { membench_launch<8, 1, 1, 7, 4>(u0, u1, fp, 0); return true; }
      return false;
   };
   if (!launch()) { probe_err("membench: unsupported (R,WY,PF)"); return -1.0; }
   hipDeviceSynchronize();
   hipEventRecord(e0, 0);
   for (int i = 0; i < reps; i++) launch();
   hipEventRecord(e1, 0);
   hipEventSynchronize(e1);
   float ms = 0;
   hipEventElapsedTime(&ms, e0, e1);
   hipEventDestroy(e0); hipEventDestroy(e1);
   if (hipGetLastError() != hipSuccess) { probe_err("membench launch failed"); return -1.0; }
   return ms / reps;
}

double pf_tb2_probe(const void *A, const void *B, void *C, void *D, int64_t Nx, int64_t Ny, int64_t Nz, double a1, double a2,
                    int32_t margin, int32_t tye, int32_t chunk, int32_t reps) {
   pf::Tb2Params tp{};
   const int64_t P = grid_pitch(Nz, 4);
   tp.A = (const float *)A; tp.B = (const float *)B; tp.C = (float *)C; tp.D = (float *)D;
   tp.plane = Ny * P; tp.Nx = (int)Nx; tp.Ny = (int)Ny; tp.Nz = (int)Nz; tp.P = (int)P;
   if (margin < 2 || (margin % 4) != 0 || ((Nz - 2 * margin) % 4) != 0) { probe_err("tb2 probe: margin must be a multiple of 4 >= 4"); return -1.0; }
   tp.x_begin = margin; tp.x_end = (int)Nx - margin;
   tp.y_begin = margin; tp.z_begin = margin;
   tp.chunk = chunk > 0 ? chunk : 64;
   tp.nxc = (int)cdiv(tp.x_end - tp.x_begin, tp.chunk);
   tp.nzt = (int)cdiv(Nz - 2 * margin, 248);
   hipEvent_t e0, e1;
   hipEventCreate(&e0); hipEventCreate(&e1);
   // +100000: A,B (and C,D) interleaved row by row in one storage (the caller passes B = A + P, D = C + P); +200000: plane by plane
   // (B = A + Ny*P).  Layout experiments for the number of concurrent DRAM streams; the column clamp is then off by a row, which
   // only matters for the bits of the last columns, not for the time.
   const int lay = tye / 100000; tye %= 100000;
   if (lay == 1) { tp.P = (int)(2 * P); tp.plane = 2 * Ny * P; }
   if (lay == 2) tp.plane = 2 * Ny * P;
   tp.band = tye >= 10000 ? 1 : 0; // +10000: banded tile order
   if (tye >= 10000) tye -= 10000;
   auto nblk = [&]() { const uint32_t T = (uint32_t)tp.nzt * tp.nyt; return tp.band ? 8 * ((T + 7) / 8) * (uint32_t)tp.nxc : T * (uint32_t)tp.nxc; };
   auto launch = [&]() {
      if (tye == 20) { tp.nyt = (int)cdiv(Ny - 2 * margin, 16); hipLaunchKernelGGL((pf::k_tb2_proto<20, 8>), dim3((uint32_t)tp.nzt * tp.nyt * tp.nxc), dim3(512), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 12) { tp.nyt = (int)cdiv(Ny - 2 * margin, 8); hipLaunchKernelGGL((pf::k_tb2_proto<12, 4>), dim3((uint32_t)tp.nzt * tp.nyt * tp.nxc), dim3(256), 0, 0, tp, (float)a1, (float)a2); return true; }
      // register-resident variants: tye = 100*R + WY
      if (tye == 204) { tp.nyt = (int)cdiv(Ny - 2 * margin, 8); hipLaunchKernelGGL((pf::k_tb2_reg<float, 2, 4>), dim3((uint32_t)tp.nzt * tp.nyt * tp.nxc), dim3(256), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 202) { tp.nyt = (int)cdiv(Ny - 2 * margin, 4); hipLaunchKernelGGL((pf::k_tb2_reg<float, 2, 2>), dim3((uint32_t)tp.nzt * tp.nyt * tp.nxc), dim3(128), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 104) { tp.nyt = (int)cdiv(Ny - 2 * margin, 4); hipLaunchKernelGGL((pf::k_tb2_reg<float, 1, 4>), dim3((uint32_t)tp.nzt * tp.nyt * tp.nxc), dim3(256), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 304) { tp.nyt = (int)cdiv(Ny - 2 * margin, 12); hipLaunchKernelGGL((pf::k_tb2_reg<float, 3, 4>), dim3((uint32_t)tp.nzt * tp.nyt * tp.nxc), dim3(256), 0, 0, tp, (float)a1, (float)a2); return true; }
      // +2000: halo rows through LDS (k_tb2_lds)
      if (tye == 2304) { tp.nyt = (int)cdiv(Ny - 2 * margin, 12); hipLaunchKernelGGL((pf::k_tb2_lds<3, 4>), dim3(nblk()), dim3(256), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 2308) { tp.nyt = (int)cdiv(Ny - 2 * margin, 24); hipLaunchKernelGGL((pf::k_tb2_lds<3, 8>), dim3(nblk()), dim3(512), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 2404) { tp.nyt = (int)cdiv(Ny - 2 * margin, 16); hipLaunchKernelGGL((pf::k_tb2_lds<4, 4>), dim3(nblk()), dim3(256), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 2204) { tp.nyt = (int)cdiv(Ny - 2 * margin, 8); hipLaunchKernelGGL((pf::k_tb2_lds<2, 4>), dim3(nblk()), dim3(256), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 2208) { tp.nyt = (int)cdiv(Ny - 2 * margin, 16); hipLaunchKernelGGL((pf::k_tb2_lds<2, 8>), dim3(nblk()), dim3(512), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 1316) { tp.nyt = (int)cdiv(Ny - 2 * margin, 48); hipLaunchKernelGGL((pf::k_tb2_reg<float, 3, 16, false>), dim3(nblk()), dim3(1024), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 1216) { tp.nyt = (int)cdiv(Ny - 2 * margin, 32); hipLaunchKernelGGL((pf::k_tb2_reg<float, 2, 16, false>), dim3(nblk()), dim3(1024), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 1208) { tp.nyt = (int)cdiv(Ny - 2 * margin, 16); hipLaunchKernelGGL((pf::k_tb2_reg<float, 2, 8, false>), dim3(nblk()), dim3(512), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 1404) { tp.nyt = (int)cdiv(Ny - 2 * margin, 16); hipLaunchKernelGGL((pf::k_tb2_reg<float, 4, 4, false>), dim3(nblk()), dim3(256), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 1408) { tp.nyt = (int)cdiv(Ny - 2 * margin, 32); hipLaunchKernelGGL((pf::k_tb2_reg<float, 4, 8, false>), dim3(nblk()), dim3(512), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 1308) { tp.nyt = (int)cdiv(Ny - 2 * margin, 24); hipLaunchKernelGGL((pf::k_tb2_reg<float, 3, 8, false>), dim3(nblk()), dim3(512), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 1304) { tp.nyt = (int)cdiv(Ny - 2 * margin, 12); hipLaunchKernelGGL((pf::k_tb2_reg<float, 3, 4, false>), dim3(nblk()), dim3(256), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 308) { tp.nyt = (int)cdiv(Ny - 2 * margin, 24); hipLaunchKernelGGL((pf::k_tb2_reg<float, 3, 8>), dim3((uint32_t)tp.nzt * tp.nyt * tp.nxc), dim3(512), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 302) { tp.nyt = (int)cdiv(Ny - 2 * margin, 6); hipLaunchKernelGGL((pf::k_tb2_reg<float, 3, 2>), dim3((uint32_t)tp.nzt * tp.nyt * tp.nxc), dim3(128), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 408) { tp.nyt = (int)cdiv(Ny - 2 * margin, 32); hipLaunchKernelGGL((pf::k_tb2_reg<float, 4, 8>), dim3((uint32_t)tp.nzt * tp.nyt * tp.nxc), dim3(512), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 404) { tp.nyt = (int)cdiv(Ny - 2 * margin, 16); hipLaunchKernelGGL((pf::k_tb2_reg<float, 4, 4>), dim3((uint32_t)tp.nzt * tp.nyt * tp.nxc), dim3(256), 0, 0, tp, (float)a1, (float)a2); return true; }
      if (tye == 24) { tp.nyt = (int)cdiv(Ny - 2 * margin, 20); hipLaunchKernelGGL((pf::k_tb2_proto<24, 8>), dim3((uint32_t)tp.nzt * tp.nyt * tp.nxc), dim3(512), 0, 0, tp, (float)a1, (float)a2); return true; }
      return false;
   };
   if (!launch()) { probe_err("tb2 probe: tye must be 12, 20, 24 (LDS) or 100*R+WY (registers)"); return -1.0; }
   if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { probe_err("tb2 probe launch failed"); return -1.0; }
   hipEventRecord(e0, 0);
   for (int i = 0; i < reps; i++) launch();
   hipEventRecord(e1, 0);
   hipEventSynchronize(e1);
   float ms = 0;
   hipEventElapsedTime(&ms, e0, e1);
   hipEventDestroy(e0); hipEventDestroy(e1);
   return reps > 0 ? ms / reps : 0.0;
}


// three fused steps (the product kernel k_tb3, pf_tb3.h): A = u^{n-1}, B = u^n -> D = u^{n+2}, E = u^{n+3} on the box [m, N-m)^3.  variant = 100*R + WT
// (+10000: banded tile order).  Returns the average milliseconds per launch (<0 on error).
double pf_tb3_probe(const void *A, const void *B, void *D, void *E, int64_t Nx, int64_t Ny, int64_t Nz, double a1, double a2,
                    int32_t margin, int32_t variant, int32_t chunk, int32_t reps) {
   pf::Tb2Params tp{};
   const int64_t P = grid_pitch(Nz, 4);
   tp.A = (const float *)A; tp.B = (const float *)B; tp.C = nullptr; tp.D = (float *)D; tp.E = E;
   tp.plane = Ny * P; tp.Nx = (int)Nx; tp.Ny = (int)Ny; tp.Nz = (int)Nz; tp.P = (int)P;
   if (margin < 4 || (margin % 4) != 0 || ((Nz - 2 * margin) % 4) != 0) { probe_err("tb3 probe: margin must be a multiple of 4 >= 4"); return -1.0; }
   tp.x_begin = margin; tp.x_end = (int)Nx - margin;
   tp.y_begin = margin; tp.z_begin = margin;
   tp.chunk = chunk > 0 ? chunk : 32;
   tp.nxc = (int)cdiv(tp.x_end - tp.x_begin, tp.chunk);
   tp.nzt = (int)cdiv(Nz - 2 * margin, 248);
   tp.band = variant >= 10000 ? 1 : 0;
   variant %= 10000;
   auto nblk = [&]() { const uint32_t T = (uint32_t)tp.nzt * tp.nyt; return tp.band ? 8 * ((T + 7) / 8) * (uint32_t)tp.nxc : T * (uint32_t)tp.nxc; };
   auto launch = [&]() {
#define PF_TB3(r, wt) if (variant == 100 * r + wt) { tp.nyt = (int)cdiv(Ny - 2 * margin, wt * r - 4); hipLaunchKernelGGL((pf::k_tb3<float, r, wt>), dim3(nblk()), dim3(64 * wt), 0, 0, tp, (float)a1, (float)a2); return true; }
      PF_TB3(3, 8) PF_TB3(2, 8) PF_TB3(3, 4) PF_TB3(4, 4) PF_TB3(2, 12) PF_TB3(3, 6)
#undef PF_TB3
      return false;
   };
   hipEvent_t e0, e1;
   hipEventCreate(&e0); hipEventCreate(&e1);
   if (!launch()) { probe_err("tb3 probe: variant = 100*R + WT, one of 308 208 304 404 212 306"); return -1.0; }
   if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { probe_err("tb3 probe launch failed"); return -1.0; }
   hipEventRecord(e0, 0);
   for (int i = 0; i < reps; i++) launch();
   hipEventRecord(e1, 0);
   hipEventSynchronize(e1);
   float ms = 0;
   hipEventElapsedTime(&ms, e0, e1);
   hipEventDestroy(e0); hipEventDestroy(e1);
   return reps > 0 ? ms / reps : 0.0;
}

} // extern "C"
