// tools/gather_roof.hip -- what the MEMORY side of the config-5 AdvectionRK45 kernel can attain on this GPU: the kernel's access pattern
// without its arithmetic.  Not part of the product; built by tools/build_gather_roof.sh, run by bench.py's
// `secondary` leg when the binary is there (`roofline.attainable`).
//
// The pattern (profiles/r04_r_rk45_search_outcomes.txt): one lane per particle in one-wavefront workgroups at 3 waves
// per SIMD (168 registers, 12.3 KB of LDS per workgroup), particles sorted by cell.  A lane reads its 88 B of state, makes A attempts of six
// evaluations each and writes 88 + 8 B back.  An evaluation whose sample point left the cached cell (probability P, drawn per lane and
// evaluation; 0.141 measured) fetches, DEPENDENT on the previous one (the new cell index comes out of the data just loaded): the two
// 128-byte lines of the neighbour cell's record in the 256 B / cell table (3.4 GB at 4322 x 3059) and the four lines that hold its
// staggered U0, U1, V0, V1 on two time levels of the packed {U, V, W} float32 levels (2 x 11.9 GB) -- six lines per cell change.  Every
// wave executes that path in lock step for the lanes that need it (exec-masked loads), like the kernel does.
//
//   gather_roof [--particles 1e7] [--attempts 5] [--p 0.141] [--alu 0] [--reps 5] [--nx 4322 --ny 3059 --nz 75]
// prints one JSON object: ms per launch (median), the algorithmic GB/s at SURVEY 8(d)'s 672 B per attempt, lines fetched per second.
// --alu K puts K dependent fp64 FMAs between two evaluations (0 = the pure memory roof).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                           \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)

struct Args {
    const double* ct2;   // [ncell][32]
    const float* lev0;   // [nz][ny][nx][3]
    const float* lev1;
    const double* state_in;  // 11 columns of n
    double* state_out;       // 12 columns of n
    const int32_t* cell;     // j * nx + i of the lane's first cell
    const int32_t* depth;    // its depth level
    int64_t n;
    int nx, ny, nz, attempts, alu;
    uint32_t p_thresh;       // P * 2^32
};

__device__ __forceinline__ uint32_t mix(uint32_t a, uint32_t b) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0xC2B2AE3Du;
    h ^= h >> 13;
    return h;
}

__global__ void __launch_bounds__(64, 3) gather_kernel(const Args a) {
    extern __shared__ double smem[];
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n) return;
    double acc = 0.0;
    for (int c = 0; c < 11; c++) acc += a.state_in[(int64_t)c * a.n + i];
    int cell = a.cell[i];
    const int kz = a.depth[i];
    const int ncell = a.nx * a.ny;
    double* slot = smem + threadIdx.x;  // the lane's LDS slot (15 record rows: written on a cell change, read by every evaluation)
    for (int k = 0; k < 15; k++) slot[k * 64] = 0.0;
    for (int at = 0; at < a.attempts; at++) {
        for (int ev = 0; ev < 6; ev++) {
            const uint32_t h = mix((uint32_t)i, (uint32_t)(at * 6 + ev));
            if (h < a.p_thresh) {  // the sample point left the cell: the neighbour the previous data points at
                const int dir = (int)((h >> 3) & 3u);
                int nc = cell + (dir == 0 ? 1 : dir == 1 ? -1 : dir == 2 ? a.nx : -a.nx) + (int)(acc * 0.0);  // (depends on what was loaded)
                nc = nc < a.nx ? nc + a.nx : (nc >= ncell - a.nx - 1 ? nc - a.nx - 1 : nc);
                cell = nc;
                const double2* r = reinterpret_cast<const double2*>(a.ct2 + (int64_t)cell * 32);
                double2 v[12];
#pragma unroll
                for (int k = 0; k < 12; k++) v[k] = r[k];  // 192 B: both lines of the record
                const int64_t e = ((int64_t)kz * ncell + cell) * 3;
                const float f0 = a.lev0[e], f1 = a.lev0[e + 3], f2 = a.lev0[e + 1], f3 = a.lev0[e + (int64_t)a.nx * 3 + 1];
                const float g0 = a.lev1[e], g1 = a.lev1[e + 3], g2 = a.lev1[e + 1], g3 = a.lev1[e + (int64_t)a.nx * 3 + 1];
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    slot[(2 * k) * 64] = v[k].x;
                    slot[(2 * k + 1) * 64] = v[k].y;
                }
                slot[14 * 64] = v[7].x;
                acc += v[8].x + v[9].y + v[10].x + v[11].y + (double)(f0 + f1 + f2 + f3) + (double)(g0 + g1 + g2 + g3);
            }
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 15; k++) s += slot[k * 64];  // the point-in-cell test reads the 15 rows
            acc += s * 1e-300;
            for (int k = 0; k < a.alu; k++) acc = __builtin_fma(acc, 1.0000000001, 1e-30);  // stand-in for the arithmetic (--alu)
        }
    }
    for (int c = 0; c < 12; c++) a.state_out[(int64_t)c * a.n + i] = acc + c;
}

__global__ void fill_kernel(float* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (float)(i & 1023) * 1e-3f;
}

int main(int argc, char** argv) {
    double particles = 1e7, p = 0.141;
    int attempts = 5, alu = 0, reps = 5, nx = 4322, ny = 3059, nz = 75;
    for (int k = 1; k + 1 < argc; k += 2) {
        const std::string o = argv[k];
        const char* v = argv[k + 1];
        if (o == "--particles") particles = atof(v);
        else if (o == "--attempts") attempts = atoi(v);
        else if (o == "--p") p = atof(v);
        else if (o == "--alu") alu = atoi(v);
        else if (o == "--reps") reps = atoi(v);
        else if (o == "--nx") nx = atoi(v);
        else if (o == "--ny") ny = atoi(v);
        else if (o == "--nz") nz = atoi(v);
    }
    const int64_t n = (int64_t)particles, ncell = (int64_t)nx * ny;
    Args a{};
    double* ct2;
    float *l0, *l1;
    const size_t ct2_b = (size_t)ncell * 32 * 8, lev_b = (size_t)ncell * nz * 3 * 4;
    CK(hipMalloc(&ct2, ct2_b));
    CK(hipMalloc(&l0, lev_b));
    CK(hipMalloc(&l1, lev_b));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (float*)ct2, (int64_t)(ct2_b / 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, l0, (int64_t)(lev_b / 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, l1, (int64_t)(lev_b / 4));
    // particles: uniform over the interior cells and over 60 of the depth levels, sorted by (cell, depth) like the product's cell sort
    std::vector<uint64_t> key((size_t)n);
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int ci = (int)(0.1 * nx + (s % 1000003) / 1000003.0 * 0.8 * nx), cj = (int)(0.1 * ny + ((s >> 20) % 1000003) / 1000003.0 * 0.8 * ny);
        const int kz = (int)((s >> 40) % (uint64_t)std::max(1, nz * 4 / 5));
        key[i] = ((uint64_t)(cj * (int64_t)nx + ci) << 8) | (uint64_t)kz;
    }
    std::sort(key.begin(), key.end());
    std::vector<int32_t> cell((size_t)n), depth((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        cell[i] = (int32_t)(key[i] >> 8);
        depth[i] = (int32_t)(key[i] & 255);
    }
    int32_t *d_cell, *d_depth;
    double *sin_, *sout;
    CK(hipMalloc(&d_cell, n * 4));
    CK(hipMalloc(&d_depth, n * 4));
    CK(hipMalloc(&sin_, n * 8 * 11));
    CK(hipMalloc(&sout, n * 8 * 12));
    CK(hipMemcpy(d_cell, cell.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_depth, depth.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(sin_, 0, n * 8 * 11));
    a.ct2 = ct2; a.lev0 = l0; a.lev1 = l1; a.state_in = sin_; a.state_out = sout; a.cell = d_cell; a.depth = d_depth;
    a.n = n; a.nx = nx; a.ny = ny; a.nz = nz; a.attempts = attempts; a.alu = alu;
    a.p_thresh = (uint32_t)(p * 4294967296.0);
    const size_t lds = 13600;  // 12 workgroups per CU = 3 waves per SIMD: the residency of the product kernel (168 VGPRs, 12.3 KB of LDS)
    int wg_per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, gather_kernel, 64, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < reps + 1; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), lds, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r) ms.push_back(t);  // (the first launch is cold)
    }
    std::sort(ms.begin(), ms.end());
    const double med = ms[(ms.size() - 1) / 2];
    const double units = (double)n * attempts, changes = units * 6 * p;
    printf("{\"what\": \"access pattern of the config-5 AdvectionRK45 kernel without its arithmetic (tools/gather_roof.hip)\", \"particles\": %lld, "
           "\"attempts_per_particle\": %d, \"cell_change_probability\": %.3f, \"alu_fmas_per_evaluation\": %d, \"workgroups_per_cu\": %d, "
           "\"ms_per_launch_median\": %.4f, \"ms_min\": %.4f, \"ms_max\": %.4f, \"attempts\": %.0f, \"algorithmic_bytes_per_attempt\": 672, "
           "\"attainable_GBps_algorithmic\": %.1f, \"frac_of_8TBps\": %.4f, \"lines_128B_per_s\": %.4g, \"line_GBps\": %.1f, "
           "\"ms_for_5.034e7_attempts\": %.3f}\n",
           (long long)n, attempts, p, alu, wg_per_cu, med, ms.front(), ms.back(), units, 672.0 * units / (med * 1e-3) / 1e9,
           672.0 * units / (med * 1e-3) / 1e9 / 8000.0, changes * 6 / (med * 1e-3), changes * 6 * 128 / (med * 1e-3) / 1e9, med * 5.034e7 / units);
    return 0;
}
