// Instantiations of the fused advection kernel for program PROG_RK4_3D (one TU per program: parallel build).
// 4 waves per SIMD (128 VGPRs): measured on MI355X, C2: 8.4e9 (2 waves) / 1.15e10 (3) / 1.24e10 (4) particle-steps/s. The kernel is VALU-issue bound and
// needs the extra wave to cover gather latency; the few spills this forces cost less than the lost occupancy)
#ifndef PK_MIN_WAVES
#define PK_MIN_WAVES 4
#endif
// curvilinear / C-grid instantiations: 3 waves per SIMD (168 VGPRs, ~20 spilled) beat 2 on the NEMO-size grid (66.4 vs 69.4 ms per
// 2.4e8 particle-steps): the kernel waits on memory for a third of its wave cycles, the third wave covers part of it
#ifndef PK_MIN_WAVES_HEAVY
#define PK_MIN_WAVES_HEAVY 3
#endif
#include "pk_kernels.h"
namespace pk {
PK_DEFINE_LAUNCH_PROGRAM(PROG_RK4_3D, PK_KERNEL_ADVECTION_RK4_3D, 0, false)
}
