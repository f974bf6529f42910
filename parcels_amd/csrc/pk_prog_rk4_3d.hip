// Instantiations of the fused advection kernel for program PROG_RK4_3D (one TU per program: parallel build).
#include "pk_kernels.h"
namespace pk {
PK_DEFINE_LAUNCH_PROGRAM(PROG_RK4_3D, PK_KERNEL_ADVECTION_RK4_3D, 0)
}
