// Fast instantiations of the fused advection kernel for AdvectionRK4 on a rectilinear A-grid with float64 coordinates
// (pk_fast_agrid.h; BASELINE config 2), field dtype x particle dtype.
#ifndef PK_MIN_WAVES
#define PK_MIN_WAVES 4
#endif
#include "pk_kernels.h"
namespace pk {
PK_DEFINE_LAUNCH_FAST(PROG_RK4, PK_KERNEL_ADVECTION_RK4)
}
