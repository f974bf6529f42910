// Instantiations of the fused advection kernel for AdvectionRK4 / AdvectionRK4_3D with CGrid_Velocity on a spherical curvilinear
// C-grid with float64 node coordinates (pk_fast_cgrid.h): field dtype x particle dtype x 2-D / 3-D x (every cell of the grid spans less
// than 2^-8 rad of latitude: FastC::near_edges, the edge cosines of CGrid_Velocity from the sample's own -- cos_near).
#include "pk_kernels.h"
namespace pk {
#define PK_CG_CASE(FT, PF, D3V)                                                                                               \
    do {                                                                                                                      \
        if (print_occupancy()) {                                                                                              \
            int nb = 0;                                                                                                       \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, advect_cgrid_kernel<FT, PF, D3V, false>, FC_LANES, lds_bytes); \
            fprintf(stderr, "[pk] advect_cgrid_kernel<%s,pf %d,d3 %d> wg %d lds %zu B: %d workgroups / CU\n",                \
                    sizeof(FT) == 4 ? "f32" : "f64", PF, (int)D3V, FC_LANES, (size_t)lds_bytes, nb);                          \
        }                                                                                                                     \
        if (a.fastc.near_edges) hipLaunchKernelGGL((advect_cgrid_kernel<FT, PF, D3V, true>), grid, dim3(FC_LANES), lds_bytes, stream, a); \
        else hipLaunchKernelGGL((advect_cgrid_kernel<FT, PF, D3V, false>), grid, dim3(FC_LANES), lds_bytes, stream, a);     \
    } while (0)
void launch_cgrid(int field_f32, int particles_f32, int d3, const KArgs& a, int64_t n, size_t lds_bytes, hipStream_t stream) {
    const dim3 grid((unsigned)((n + FC_LANES - 1) / FC_LANES));
    const int key = (field_f32 ? 4 : 0) | (particles_f32 ? 2 : 0) | (d3 ? 1 : 0);
    switch (key) {
        case 0: PK_CG_CASE(double, 0, false); break;
        case 1: PK_CG_CASE(double, 0, true); break;
        case 2: PK_CG_CASE(double, 1, false); break;
        case 3: PK_CG_CASE(double, 1, true); break;
        case 4: PK_CG_CASE(float, 0, false); break;
        case 5: PK_CG_CASE(float, 0, true); break;
        case 6: PK_CG_CASE(float, 1, false); break;
        default: PK_CG_CASE(float, 1, true); break;
    }
}
#define PK_CG_NE(K, FT, PF)                                                                                     \
    do {                                                                                                        \
        if (a.fastc.near_edges) hipLaunchKernelGGL((K<FT, PF, true>), grid, dim3(FC_LANES), lds_bytes, stream, a); \
        else hipLaunchKernelGGL((K<FT, PF, false>), grid, dim3(FC_LANES), lds_bytes, stream, a);                 \
    } while (0)
void launch_cgrid_rk45(int field_f32, int particles_f32, const KArgs& a, int64_t n, size_t lds_bytes, hipStream_t stream) {
    const dim3 grid((unsigned)((n + FC_LANES - 1) / FC_LANES));
    if (field_f32) {
        if (particles_f32) PK_CG_NE(advect_cgrid_rk45_kernel, float, 1);
        else PK_CG_NE(advect_cgrid_rk45_kernel, float, 0);
    } else {
        if (particles_f32) PK_CG_NE(advect_cgrid_rk45_kernel, double, 1);
        else PK_CG_NE(advect_cgrid_rk45_kernel, double, 0);
    }
}
void launch_cgrid_m1(int field_f32, int particles_f32, const KArgs& a, int64_t n, size_t lds_bytes, hipStream_t stream) {
    const dim3 grid((unsigned)((n + FC_LANES - 1) / FC_LANES));
    if (field_f32) {
        if (particles_f32) PK_CG_NE(advect_cgrid_m1_kernel, float, 1);
        else PK_CG_NE(advect_cgrid_m1_kernel, float, 0);
    } else {
        if (particles_f32) PK_CG_NE(advect_cgrid_m1_kernel, double, 1);
        else PK_CG_NE(advect_cgrid_m1_kernel, double, 0);
    }
}
}  // namespace pk
