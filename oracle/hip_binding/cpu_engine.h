/* Test infrastructure (oracle/Makefile target `hipbind`): found ahead of the reference's own c_cuda/cpu_engine.h by
 * include order, so that the reference's UNMODIFIED c_cuda/fdtd_main.c -- which picks its engine with
 * `#include <cpu_engine.h>` at fdtd_main.c:29-33 -- is compiled against the binding of INTEGRATION.md section 2.
 * It stands for the `#elif USING_HIP / #include <hip_engine.h>` branch a maintainer would add there. */
#include "hip_engine.h"
