/*
 * pf_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU oracle for the FDTD time-step hot path: restates c_cuda/cpu_engine.h (run_sim :52-360,
 * process_bnl_pts_fd :363-405) of the reference for Real=float and Real=double in one library.
 * Parity is PINNED: tests/test_oracle_pinned.py checks this file bit-for-bit against the compiled
 * reference (oracle/_ref/fdtd_main_cpu_{single,double}.x, built by oracle/Makefile from the sources
 * where they lie under /root/reference) and against the golden vectors under tests/golden/ that
 * were captured from those binaries.
 *
 * A second mode (orc_set_numerics / oracle_run_sim_num with safeguarded = 1) restates the arithmetic of the reference's CUDA
 * engine (c_cuda/gpu_engine.h:220-274,288-365); that mode is PARITY-UNPINNED (no nvcc here), see pf_oracle_impl.inc.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Build: see oracle/Makefile (-O3 -fopenmp -ffp-contract=off, baseline x86-64 like the reference Makefile).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <fenv.h>
#include <math.h>
#include <omp.h>
#include "pffdtd_hip.h"

#define REAL float
#define SFX _f32
#include "pf_oracle_impl.inc"
#undef REAL
#undef SFX

#define REAL double
#define SFX _f64
#include "pf_oracle_impl.inc"
#undef REAL
#undef SFX

int oracle_max_threads(void) { return omp_get_max_threads(); }
void oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
