// Instantiations of the fused advection kernel for program PROG_RK4 (one TU per program: parallel build).
// 3 waves per SIMD: measured +25% on MI355X over the unconstrained allocation (the kernel is VALU-issue bound and
// needs the extra wave to cover gather latency; the few spills this forces cost less than the lost occupancy)
#define PK_MIN_WAVES 3
#include "pk_kernels.h"
namespace pk {
PK_DEFINE_LAUNCH_PROGRAM(PROG_RK4, PK_KERNEL_ADVECTION_RK4, 0)
}
