/* pf_probe.h -- C ABI of tools/libpf_probe.so: calibration and research probes used by tools/membench.py and
 * tools/tb2_probe.py.  Deliberately NOT part of include/pffdtd_hip.h (the product ABI). */
#ifndef PF_PROBE_H
#define PF_PROBE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
const char *pf_probe_last_error(void);
/* ---- calibration (tools/membench.py): time `reps` launches of a streaming kernel over two float grids of
 * Nx*Ny*P elements; kind 0 = linear stream, 1 = the marching tile pattern (R rows/lane, WY waves, prefetch PF planes).
 * Returns the average milliseconds per launch (<0 on error). */
double pf_membench(void *u0, void *u1, int64_t Nx, int64_t Ny, int64_t Nz, int32_t kind, int32_t R, int32_t WY,
                   int32_t PF, int32_t chunk, int32_t swizzle, int32_t reps);

/* ---- research probe (tools/tb2_probe.py): two fused leap-frog steps of the pure 7-point air update on the box
 * [m, N-m)^3 of float grids A=u^{n-1}, B=u^n -> C=u^{n+1}, D=u^{n+2} (padded layout, pf_grid_pitch).  Not used by the
 * engine.  Returns the average milliseconds per launch (<0 on error). */
double pf_tb2_probe(const void *A, const void *B, void *C, void *D, int64_t Nx, int64_t Ny, int64_t Nz, double a1, double a2,
                    int32_t margin, int32_t tye, int32_t chunk, int32_t reps);

/* ---- research probe (tools/tb3_probe.py): THREE fused steps, A=u^{n-1}, B=u^n -> D=u^{n+2}, E=u^{n+3} (k_tb3, pf_tb3.h).
 * variant = 100*R + WT (+10000: banded tile order).  Returns the average milliseconds per launch (<0 on error). */
double pf_tb3_probe(const void *A, const void *B, void *D, void *E, int64_t Nx, int64_t Ny, int64_t Nz, double a1, double a2,
                    int32_t margin, int32_t variant, int32_t chunk, int32_t reps);

#ifdef __cplusplus
}
#endif
#endif
