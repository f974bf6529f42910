// Instantiations of the fused advection kernel for program PROG_RK45 (BASELINE config 5: the divergent-dt path).
#ifndef PK_MIN_WAVES
#define PK_MIN_WAVES 2
#endif
#include "pk_kernels.h"
namespace pk {
PK_DEFINE_LAUNCH_PROGRAM(PROG_RK45, PK_KERNEL_ADVECTION_RK45, 0, false)
}
