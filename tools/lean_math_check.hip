// tools/lean_math_check.hip -- accuracy of sqrt_lean / div_lean / rcp_lean / sincos_near (parcels_amd/csrc/pk_fast_cgrid.h) on the device,
// against the correctly rounded library routines over random operands of the magnitudes the C-grid evaluation feeds them.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I parcels_amd/csrc tools/lean_math_check.hip -o /tmp/lean_math_check && /tmp/lean_math_check
// (tests/test_gpu_lean_math.py does that on the GPU box)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "pk_fast_cgrid.h"  // the routines under test, as shipped (-I parcels_amd/csrc)

__device__ uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__device__ double uni(uint64_t& s) { return (double)(rng(s) >> 11) * (1.0 / 9007199254740992.0); }
__device__ double ulps(double got, double want) {
    if (got == want) return 0.0;
    int e;
    frexp(want, &e);
    return fabs(got - want) / ldexp(1.0, e - 53);
}

// out[0..3]: max ulp error of sqrt, div, x * rcp (vs x / b), and max ABSOLUTE error of sin / cos near (vs the library sincos of a0 + d) in units of 2^-53
__global__ void check(double* out, int per) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    double m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    for (int k = 0; k < per; k++) {
        const double x = exp(uni(s) * 60.0 - 20.0);  // 2e-9 .. 2e17
        m0 = fmax(m0, ulps(pk::sqrt_lean(x), sqrt(x)));
        const double a = (uni(s) - 0.5) * exp(uni(s) * 40.0 - 20.0), b = (uni(s) < 0.5 ? -1.0 : 1.0) * exp(uni(s) * 60.0 - 30.0);
        m1 = fmax(m1, ulps(pk::div_lean(a, b), a / b));
        m2 = fmax(m2, ulps(a * pk::rcp_lean(b), a / b));
        const double a0 = (uni(s) - 0.5) * 6.4, d = (uni(s) - 0.5) * 0.015625;
        double s0, c0, sn, cn, se, ce;
        sincos(a0, &s0, &c0);
        pk::sincos_near(d, s0, c0, sn, cn);
        sincos(a0 + d, &se, &ce);  // (a0 + d rounds: an angle error of 2^-53 * |a0| at most -- counted against the near routine here)
        m3 = fmax(m3, fmax(fabs(sn - se), fabs(cn - ce)) * 9007199254740992.0);
    }
    atomicMax((unsigned long long*)&out[0], (unsigned long long)__double_as_longlong(m0));
    atomicMax((unsigned long long*)&out[1], (unsigned long long)__double_as_longlong(m1));
    atomicMax((unsigned long long*)&out[2], (unsigned long long)__double_as_longlong(m2));
    atomicMax((unsigned long long*)&out[3], (unsigned long long)__double_as_longlong(m3));
}

int main() {
    double* d;
    if (hipMalloc(&d, 4 * sizeof(double)) != hipSuccess) return 2;
    hipMemset(d, 0, 4 * sizeof(double));
    hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, 0, d, 400);
    double h[4];
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("{\"samples\": %d, \"sqrt_lean_max_ulp\": %.3f, \"div_lean_max_ulp\": %.3f, \"mul_rcp_lean_max_ulp\": %.3f, \"sincos_near_max_abs_err_in_2^-53\": %.3f}\n",
           1024 * 256 * 400, h[0], h[1], h[2], h[3]);
    return (h[0] <= 1.0 && h[1] <= 1.0 && h[2] <= 2.0 && h[3] <= 4.0) ? 0 : 1;
}
