// Instantiations of the fused advection kernel for program PROG_M1 (BASELINE config 5: the divergent-dt path).
#ifndef PK_MIN_WAVES
#define PK_MIN_WAVES 2
#endif
#include "pk_kernels.h"
namespace pk {
PK_DEFINE_LAUNCH_PROGRAM(PROG_M1, PK_KERNEL_ADVECTIONDIFFUSION_M1, 0, false)
}
