// Instantiations of the fused advection kernel for program PROG_TYPED: the kernel-list interpreter with NumPy's float32 dtype
// propagation compiled in (pk_device.h: TYPED) -- the program every fieldset with a float32 coordinate array runs (the
// reference's analytic datasets, BASELINE config 1).  Parity first: 2 waves per SIMD, no per-kernel specialisation.
#ifndef PK_MIN_WAVES
#define PK_MIN_WAVES 2
#endif
#include "pk_kernels.h"
namespace pk {
PK_DEFINE_LAUNCH_PROGRAM(PROG_TYPED, -1, 1, true)
}
