// pf_tb2_fcc.hip -- the 13-point two-steps-per-pass kernel (k_tb2_fcc, pf_tb2.h) in a translation unit of its own, built
// with -fno-slp-vectorize (pffdtd_amd/build.py): with the SLP vectoriser on, the compiler pairs the fp32 operations into
// v_pk_mul_f32 / v_pk_add_f32, whose aligned operand pairs cost ~200 register moves per plane and push the kernel from
// 244 registers to 348 (spills at two waves per SIMD).  The other kernels are memory-bound and keep the default flags.
#include <hip/hip_runtime.h>

#include "pf_tb2.h"

namespace pf {

// (32- and 16-lane row segments only; the 64-lane LDS-exchange variant k_tb2_fcc_x is launched from pf_engine.hip: it has
// registers to spare and gains 16 % from the packed fp32 operations this translation unit switches off)
template <typename Real> void launch_tb2_fcc(hipStream_t s, const Tb2Params &tp, Real a1, Real a2, int lw, uint32_t nblocks, bool sg, bool swz) {
   const dim3 g(nblocks), b(256);
#define PF_FCC_LW(SG, SWZ) do { if (lw == 32) hipLaunchKernelGGL((k_tb2_fcc<Real, 2, 4, 32, SG, SWZ>), g, b, 0, s, tp, a1, a2); \
                                else hipLaunchKernelGGL((k_tb2_fcc<Real, 2, 4, 16, SG, SWZ>), g, b, 0, s, tp, a1, a2); } while (0)
   if (sg) { if (swz) PF_FCC_LW(true, true); else PF_FCC_LW(true, false); }
   else { if (swz) PF_FCC_LW(false, true); else PF_FCC_LW(false, false); }
#undef PF_FCC_LW
}
template void launch_tb2_fcc<float>(hipStream_t, const Tb2Params &, float, float, int, uint32_t, bool, bool);
template void launch_tb2_fcc<double>(hipStream_t, const Tb2Params &, double, double, int, uint32_t, bool, bool);

} // namespace pf
