// Instantiations of the fused advection kernel for program PROG_GENERIC (one TU per program: parallel build).
// the kernel-list interpreter carries every built-in kernel; 2 waves per SIMD keeps it from dropping to one
#ifndef PK_MIN_WAVES
#define PK_MIN_WAVES 2
#endif
#include "pk_kernels.h"
namespace pk {
PK_DEFINE_LAUNCH_PROGRAM(PROG_GENERIC, -1, 1, false)
}
