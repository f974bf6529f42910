// Instantiations of the fused advection kernel for program PROG_GENERIC (one TU per program: parallel build).
#include "pk_kernels.h"
namespace pk {
PK_DEFINE_LAUNCH_PROGRAM(PROG_GENERIC, -1, 1)
}
