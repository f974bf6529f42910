// Instantiations of the fused advection kernel for program PROG_RK4 (one TU per program: parallel build).
#include "pk_kernels.h"
namespace pk {
PK_DEFINE_LAUNCH_PROGRAM(PROG_RK4, PK_KERNEL_ADVECTION_RK4, 0)
}
