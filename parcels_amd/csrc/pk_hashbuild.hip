// Device build of the Morton spatial hash of a curvilinear grid.
//
// What the reference does on the host with NumPy (src/parcels/_core/spatialhash.py):
//   :45-165   face bounding boxes (min/max over the 4 corner nodes) in unit-sphere Cartesian (spherical mesh) or
//             lon/lat (flat mesh); hash-grid bbox = nanmin/nanmax over the nodes
//   :214-228  bitwidth = 1023, bisected down until the total number of (hash cell, face) entries fits the budget
//             max(16 * nfaces, 2**22)
//   :269-387  every face emits one entry per hash cell its quantised box overlaps (x-major, then y, then z), entries are
//             sorted by (Morton code, face) and run-length encoded into CSR arrays keys/starts/counts/faces
//   :554-597, :647-695  bit dilation and quantisation
//
// Here: one kernel pass per bisection probe (per-face entry count + reduction), rocPRIM exclusive scan, one thread per
// entry to expand (balanced, faces that straddle many hash cells do not serialise), one 62-bit rocPRIM radix sort of the
// fused (code << 32 | face) keys, rocPRIM run-length encode.  All arithmetic that decides a quantised cell is IEEE fp64
// sub/div/mul (no contraction), the node coordinates are the ones the host computed with NumPy (node table), so the table
// is bit-identical to the reference's (tests/test_gpu_parity.py::test_device_hash_build_*).
#include "pk_hashbuild.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "pk_device.h"

namespace pk {
namespace {

constexpr int HB_BLOCK = 256;
constexpr int64_t HASH_ENTRIES_PER_FACE = 16;          // spatialhash.py:24
constexpr int64_t HASH_ENTRY_BUDGET_MIN = 1ll << 22;   // spatialhash.py:25
constexpr int HASH_MAX_BITWIDTH = 1023;                // spatialhash.py:26

struct FaceBox {  // quantised bounding box of one face: low corner and extent in hash cells
    int32_t xl, yl, zl, ny, nz;
};

__global__ void __launch_bounds__(HB_BLOCK) hb_bbox_kernel(const double* __restrict__ tab, int64_t nnodes, int spherical,
                                                           double* __restrict__ partial) {
    __shared__ double red[HB_BLOCK];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * HB_BLOCK + threadIdx.x; i < nnodes; i += (int64_t)gridDim.x * HB_BLOCK) {
        const double* r = tab + 5 * i;
        const double c[3] = {spherical ? r[2] : r[0], spherical ? r[3] : r[1], spherical ? r[4] : 0.0};
        for (int k = 0; k < 3; k++) {
            if (c[k] == c[k]) {  // nanmin / nanmax
                mn[k] = c[k] < mn[k] ? c[k] : mn[k];
                mx[k] = c[k] > mx[k] ? c[k] : mx[k];
            }
        }
    }
    for (int k = 0; k < 6; k++) {
        const bool is_min = (k & 1) == 0;
        red[threadIdx.x] = is_min ? mn[k >> 1] : mx[k >> 1];
        __syncthreads();
        for (int s = HB_BLOCK / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                const double a = red[threadIdx.x], b = red[threadIdx.x + s];
                red[threadIdx.x] = is_min ? (b < a ? b : a) : (b > a ? b : a);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * 6 + k] = red[0];
        __syncthreads();
    }
}

struct DBox {
    double v[6];
};

// Quantised box of face f; returns the number of hash cells it overlaps (0 for a face with a NaN corner).
PK_DEV int64_t face_box(const double* __restrict__ tab, int nx, int64_t f, int spherical, const DBox& bb, int bitwidth, FaceBox* out) {
    const int nfx = nx - 1;
    const int64_t j = f / nfx, i = f - j * nfx;
    const double* n00 = tab + (j * nx + i) * 5;
    const double* n10 = n00 + (int64_t)nx * 5;
    const double* nodes[4] = {n00, n00 + 5, n10 + 5, n10};
    double lo[3], hi[3];
    bool valid = true;
    for (int k = 0; k < 3; k++) {
        double l = INFINITY, h = -INFINITY;
        for (int c = 0; c < 4; c++) {
            const double v = (k == 2 && !spherical) ? 0.0 : nodes[c][spherical ? 2 + k : k];
            valid = valid && (v == v);
            l = v < l ? v : l;
            h = v > h ? v : h;
        }
        lo[k] = l;
        hi[k] = h;
    }
    if (!valid) return 0;
    const int32_t xl = (int32_t)quantize(lo[0], bb.v[0], bb.v[1], bitwidth), xh = (int32_t)quantize(hi[0], bb.v[0], bb.v[1], bitwidth);
    const int32_t yl = (int32_t)quantize(lo[1], bb.v[2], bb.v[3], bitwidth), yh = (int32_t)quantize(hi[1], bb.v[2], bb.v[3], bitwidth);
    const int32_t zl = (int32_t)quantize(lo[2], bb.v[4], bb.v[5], bitwidth), zh = (int32_t)quantize(hi[2], bb.v[4], bb.v[5], bitwidth);
    const int64_t ex = xh - xl + 1, ey = yh - yl + 1, ez = zh - zl + 1;
    if (out) {
        out->xl = xl; out->yl = yl; out->zl = zl;
        out->ny = (int32_t)ey; out->nz = (int32_t)ez;
    }
    return ex * ey * ez;
}

__global__ void __launch_bounds__(HB_BLOCK) hb_count_kernel(const double* __restrict__ tab, int nx, int64_t nfaces, int spherical, DBox bb,
                                                            int bitwidth, unsigned long long* __restrict__ total,
                                                            int64_t* __restrict__ per_face, FaceBox* __restrict__ boxes) {
    __shared__ unsigned long long red[HB_BLOCK];
    unsigned long long acc = 0;
    for (int64_t f = (int64_t)blockIdx.x * HB_BLOCK + threadIdx.x; f < nfaces; f += (int64_t)gridDim.x * HB_BLOCK) {
        FaceBox b;
        const int64_t c = face_box(tab, nx, f, spherical, bb, bitwidth, boxes ? &b : nullptr);
        if (per_face) per_face[f] = c;
        if (boxes && c > 0) boxes[f] = b;
        acc += (unsigned long long)c;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = HB_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0]) atomicAdd(total, red[0]);
}

__global__ void __launch_bounds__(HB_BLOCK) hb_expand_kernel(const int64_t* __restrict__ face_start, const FaceBox* __restrict__ boxes,
                                                             int64_t nfaces, int64_t total, unsigned long long* __restrict__ packed) {
    const int64_t e = (int64_t)blockIdx.x * HB_BLOCK + threadIdx.x;
    if (e >= total) return;
    // last face whose first entry is <= e (faces without entries share their successor's start and are skipped)
    int64_t lo = 0, hi = nfaces;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (face_start[mid] <= e) lo = mid; else hi = mid;
    }
    const FaceBox b = boxes[lo];
    const int64_t intra = e - face_start[lo];
    const int64_t nynz = (int64_t)b.ny * b.nz;
    const int64_t xi = intra / nynz, rem = intra - xi * nynz;
    const int64_t yi = rem / b.nz, zi = rem - yi * b.nz;
    const uint32_t code = (dilate_bits((uint32_t)(b.zl + zi)) << 2) | (dilate_bits((uint32_t)(b.yl + yi)) << 1) |
                          dilate_bits((uint32_t)(b.xl + xi));
    packed[e] = ((unsigned long long)code << 32) | (unsigned long long)(uint32_t)lo;
}

struct CodeOf {
    __host__ __device__ uint32_t operator()(unsigned long long v) const { return (uint32_t)(v >> 32); }
};

__global__ void __launch_bounds__(HB_BLOCK) hb_faces_kernel(const unsigned long long* __restrict__ packed, int64_t total,
                                                            uint32_t* __restrict__ faces) {
    const int64_t e = (int64_t)blockIdx.x * HB_BLOCK + threadIdx.x;
    if (e < total) faces[e] = (uint32_t)packed[e];
}

__global__ void __launch_bounds__(HB_BLOCK) hb_widen_kernel(const uint32_t* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * HB_BLOCK + threadIdx.x;
    if (k < n) out[k] = (int64_t)in[k];
}

struct Scratch {  // frees its temporaries on every exit path
    std::vector<void*> p;
    ~Scratch() {
        for (void* q : p) (void)hipFree(q);
    }
    template <class T>
    hipError_t alloc(T** out, size_t n) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) p.push_back(q);
        *out = (T*)q;
        return e;
    }
    void release(void* q) {  // the caller keeps q
        p.erase(std::remove(p.begin(), p.end(), q), p.end());
    }
    void free_now(void* q) {
        release(q);
        (void)hipFree(q);
    }
};

#define HB_TRY(call)                                     \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) {                          \
            if (err) *err = std::string(#call) + ": " + hipGetErrorString(e_); \
            return e_;                                   \
        }                                                \
    } while (0)

inline unsigned grid_for(int64_t n, int64_t cap = 1 << 16) {
    int64_t b = (n + HB_BLOCK - 1) / HB_BLOCK;
    return (unsigned)std::max<int64_t>(1, std::min(b, cap));
}

}  // namespace

namespace {
__global__ void __launch_bounds__(HB_BLOCK) hb_directory_kernel(const uint32_t* __restrict__ keys, int64_t nkeys, int shift, int64_t nbuckets,
                                                                int32_t* __restrict__ dir) {
    const int64_t b = (int64_t)blockIdx.x * HB_BLOCK + threadIdx.x;
    if (b > nbuckets) return;
    const uint64_t first = (uint64_t)b << shift;  // smallest code of bucket b (b == nbuckets: one past the last code)
    int64_t lo = 0, hi = nkeys;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((uint64_t)keys[mid] < first) lo = mid + 1; else hi = mid;
    }
    dir[b] = (int32_t)lo;
}
}  // namespace

namespace {
__global__ void __launch_bounds__(HB_BLOCK) hb_node_key_kernel(const double* __restrict__ tab, int64_t nnodes, int spherical, DBox bb, double offset,
                                                               unsigned long long* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * HB_BLOCK + threadIdx.x;
    if (i >= nnodes) return;
    const double* r = tab + 5 * i;
    unsigned long long key;
    if (spherical) {
        const double X = r[2], Y = r[3], Z = r[4];
        if (X != X || Y != Y || Z != Z) key = 0x8000000000000000ull | (unsigned long long)i;  // masked: unique
        else {
            const double s = 1048576.0;  // 2^20 cells per unit
            const unsigned long long qx = (unsigned long long)((X + 1.0) * s + offset), qy = (unsigned long long)((Y + 1.0) * s + offset),
                                     qz = (unsigned long long)((Z + 1.0) * s + offset);
            key = (qx << 42) | (qy << 21) | qz;  // each < 2^21 + 1
        }
    } else {
        const double x = r[0], y = r[1];
        if (x != x || y != y) key = 0x8000000000000000ull | (unsigned long long)i;
        else {
            const double s = 1073741824.0;  // 2^30 cells across the bbox
            const double dx = bb.v[1] - bb.v[0], dy = bb.v[3] - bb.v[2];
            const unsigned long long qx = (unsigned long long)((dx != 0 ? (x - bb.v[0]) / dx : 0.0) * s + offset),
                                     qy = (unsigned long long)((dy != 0 ? (y - bb.v[2]) / dy : 0.0) * s + offset);
            key = (qx << 31) | qy;
        }
    }
    keys[i] = key;
}

__global__ void __launch_bounds__(HB_BLOCK) hb_adjacent_equal_kernel(const unsigned long long* __restrict__ sorted, int64_t n, int* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * HB_BLOCK + threadIdx.x;
    if (i + 1 < n && sorted[i] == sorted[i + 1]) *flag = 1;
}
}  // namespace

hipError_t mesh_has_coincident_nodes(hipStream_t stream, const double* node_tab, int ny, int nx, int spherical, const double bbox[6],
                                     bool* coincident, std::string* err) {
    *coincident = true;  // the safe answer if anything fails
    const int64_t n = (int64_t)ny * nx;
    Scratch tmp;
    unsigned long long *d_keys, *d_sorted;
    int* d_flag;
    HB_TRY(tmp.alloc(&d_keys, (size_t)n));
    HB_TRY(tmp.alloc(&d_sorted, (size_t)n));
    HB_TRY(tmp.alloc(&d_flag, 1));
    HB_TRY(hipMemsetAsync(d_flag, 0, sizeof(int), stream));
    DBox bb;
    for (int k = 0; k < 6; k++) bb.v[k] = bbox[k];
    size_t tb = 0;
    void* d_tmp = nullptr;
    HB_TRY(rocprim::radix_sort_keys(nullptr, tb, d_keys, d_sorted, (size_t)n, 0u, 64u, stream));
    HB_TRY(tmp.alloc((char**)&d_tmp, tb));
    const unsigned nb = (unsigned)((n + HB_BLOCK - 1) / HB_BLOCK);
    for (int pass = 0; pass < 2; pass++) {
        hipLaunchKernelGGL(hb_node_key_kernel, dim3(nb), dim3(HB_BLOCK), 0, stream, node_tab, n, spherical, bb, pass ? 0.5 : 0.0, d_keys);
        HB_TRY(rocprim::radix_sort_keys(d_tmp, tb, d_keys, d_sorted, (size_t)n, 0u, 64u, stream));
        hipLaunchKernelGGL(hb_adjacent_equal_kernel, dim3(nb), dim3(HB_BLOCK), 0, stream, d_sorted, n, d_flag);
    }
    int flag = 1;
    HB_TRY(hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, stream));
    HB_TRY(hipStreamSynchronize(stream));
    HB_TRY(hipGetLastError());
    *coincident = flag != 0;
    return hipSuccess;
}

hipError_t build_hash_directory(hipStream_t stream, const uint32_t* keys, int64_t nkeys, int32_t** dir, int32_t* shift, std::string* err) {
    *dir = nullptr;
    *shift = 0;
    if (nkeys <= 0 || nkeys >= INT32_MAX) return hipSuccess;  // no directory: the query searches the whole array
    int bits = 1;
    while (bits < 26 && ((int64_t)1 << bits) < 2 * nkeys) bits++;  // ~0.5 keys per bucket, at most 256 MiB
    const int64_t nbuckets = (int64_t)1 << bits;
    int32_t* d = nullptr;
    HB_TRY(hipMalloc((void**)&d, (size_t)(nbuckets + 1) * sizeof(int32_t)));
    hipLaunchKernelGGL(hb_directory_kernel, dim3((unsigned)((nbuckets + 1 + HB_BLOCK - 1) / HB_BLOCK)), dim3(HB_BLOCK), 0, stream, keys, nkeys,
                       30 - bits, nbuckets, d);
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        (void)hipFree(d);
        if (err) *err = std::string("hash directory: ") + hipGetErrorString(e);
        return e;
    }
    *dir = d;
    *shift = 30 - bits;
    return hipSuccess;
}

hipError_t build_spatial_hash(hipStream_t stream, const double* node_tab, int ny, int nx, int spherical, HashBuildResult* out,
                              std::string* err) {
    *out = HashBuildResult();
    if (ny < 2 || nx < 2) {
        if (err) *err = "spatial hash needs at least 2 x 2 nodes";
        return hipErrorInvalidValue;
    }
    const int64_t nnodes = (int64_t)ny * nx, nfaces = (int64_t)(ny - 1) * (nx - 1);
    Scratch tmp;

    // 1. bbox of the hash grid
    const unsigned nb = grid_for(nnodes, 1024);
    double* d_partial;
    HB_TRY(tmp.alloc(&d_partial, (size_t)nb * 6));
    hipLaunchKernelGGL(hb_bbox_kernel, dim3(nb), dim3(HB_BLOCK), 0, stream, node_tab, nnodes, spherical, d_partial);
    std::vector<double> partial((size_t)nb * 6);
    HB_TRY(hipMemcpyAsync(partial.data(), d_partial, partial.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
    HB_TRY(hipStreamSynchronize(stream));
    DBox bb;
    for (int k = 0; k < 6; k++) {
        double v = partial[k];
        for (unsigned b = 1; b < nb; b++) v = (k & 1) ? std::max(v, partial[(size_t)b * 6 + k]) : std::min(v, partial[(size_t)b * 6 + k]);
        bb.v[k] = v;
    }
    if (!spherical) bb.v[4] = bb.v[5] = 0.0;

    // 2. bitwidth: the largest one whose entry count fits the budget (spatialhash.py:214-228)
    unsigned long long* d_total;
    HB_TRY(tmp.alloc(&d_total, 1));
    const unsigned nbf = grid_for(nfaces, 4096);
    auto total_entries = [&](int bw, int64_t* per_face, FaceBox* boxes, int64_t* result) -> hipError_t {
        hipError_t e = hipMemsetAsync(d_total, 0, sizeof(unsigned long long), stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(hb_count_kernel, dim3(nbf), dim3(HB_BLOCK), 0, stream, node_tab, nx, nfaces, spherical, bb, bw, d_total, per_face, boxes);
        unsigned long long h = 0;
        e = hipMemcpyAsync(&h, d_total, sizeof(h), hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        *result = (int64_t)h;
        return e;
    };
    const int64_t budget = std::max(HASH_ENTRIES_PER_FACE * nfaces, HASH_ENTRY_BUDGET_MIN);
    int bitwidth = HASH_MAX_BITWIDTH;
    int64_t total = 0;
    HB_TRY(total_entries(bitwidth, nullptr, nullptr, &total));
    if (total > budget) {
        int lo = 1, hi = HASH_MAX_BITWIDTH;
        while (lo < hi) {
            const int mid = (lo + hi + 1) / 2;
            HB_TRY(total_entries(mid, nullptr, nullptr, &total));
            if (total <= budget) lo = mid; else hi = mid - 1;
        }
        bitwidth = lo;
    }

    // 3. per-face entry counts and boxes at the chosen bitwidth, exclusive scan -> first entry of each face
    int64_t *d_count, *d_start;
    FaceBox* d_boxes;
    HB_TRY(tmp.alloc(&d_count, (size_t)nfaces));
    HB_TRY(tmp.alloc(&d_start, (size_t)nfaces));
    HB_TRY(tmp.alloc(&d_boxes, (size_t)nfaces));
    HB_TRY(total_entries(bitwidth, d_count, d_boxes, &total));
    if (total <= 0) {
        if (err) *err = "spatial hash: the grid has no valid face";
        return hipErrorInvalidValue;
    }
    size_t tb = 0;
    void* d_tmp = nullptr;
    HB_TRY(rocprim::exclusive_scan(nullptr, tb, d_count, d_start, (int64_t)0, (size_t)nfaces, rocprim::plus<int64_t>(), stream));
    HB_TRY(tmp.alloc((char**)&d_tmp, tb));
    HB_TRY(rocprim::exclusive_scan(d_tmp, tb, d_count, d_start, (int64_t)0, (size_t)nfaces, rocprim::plus<int64_t>(), stream));
    HB_TRY(hipStreamSynchronize(stream));
    tmp.free_now(d_tmp);
    tmp.free_now(d_count);

    // 4. expand into fused (code << 32 | face) keys and sort them
    unsigned long long *d_packed, *d_sorted;
    HB_TRY(tmp.alloc(&d_packed, (size_t)total));
    HB_TRY(tmp.alloc(&d_sorted, (size_t)total));
    hipLaunchKernelGGL(hb_expand_kernel, dim3((unsigned)((total + HB_BLOCK - 1) / HB_BLOCK)), dim3(HB_BLOCK), 0, stream, d_start, d_boxes, nfaces,
                       total, d_packed);
    tb = 0;
    HB_TRY(rocprim::radix_sort_keys(nullptr, tb, d_packed, d_sorted, (size_t)total, 0u, 62u, stream));
    HB_TRY(tmp.alloc((char**)&d_tmp, tb));
    HB_TRY(rocprim::radix_sort_keys(d_tmp, tb, d_packed, d_sorted, (size_t)total, 0u, 62u, stream));
    HB_TRY(hipStreamSynchronize(stream));
    tmp.free_now(d_tmp);
    tmp.free_now(d_packed);
    tmp.free_now(d_start);
    tmp.free_now(d_boxes);

    // 5. CSR: run-length encode the codes
    uint32_t *d_keys_full, *d_runlen, *d_faces;
    size_t* d_nruns;
    HB_TRY(tmp.alloc(&d_keys_full, (size_t)total));
    HB_TRY(tmp.alloc(&d_runlen, (size_t)total));
    HB_TRY(tmp.alloc(&d_nruns, 1));
    auto codes = rocprim::make_transform_iterator(d_sorted, CodeOf());
    tb = 0;
    HB_TRY(rocprim::run_length_encode(nullptr, tb, codes, (unsigned int)total, d_keys_full, d_runlen, d_nruns, stream));
    HB_TRY(tmp.alloc((char**)&d_tmp, tb));
    HB_TRY(rocprim::run_length_encode(d_tmp, tb, codes, (unsigned int)total, d_keys_full, d_runlen, d_nruns, stream));
    size_t nruns = 0;
    HB_TRY(hipMemcpyAsync(&nruns, d_nruns, sizeof(nruns), hipMemcpyDeviceToHost, stream));
    HB_TRY(hipStreamSynchronize(stream));
    tmp.free_now(d_tmp);
    HB_TRY(tmp.alloc(&d_faces, (size_t)total));
    hipLaunchKernelGGL(hb_faces_kernel, dim3((unsigned)((total + HB_BLOCK - 1) / HB_BLOCK)), dim3(HB_BLOCK), 0, stream, d_sorted, total, d_faces);

    uint32_t* d_keys;
    int64_t *d_counts, *d_starts;
    HB_TRY(tmp.alloc(&d_keys, nruns));
    HB_TRY(tmp.alloc(&d_counts, nruns));
    HB_TRY(tmp.alloc(&d_starts, nruns));
    HB_TRY(hipMemcpyAsync(d_keys, d_keys_full, nruns * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(hb_widen_kernel, dim3((unsigned)((nruns + HB_BLOCK - 1) / HB_BLOCK)), dim3(HB_BLOCK), 0, stream, d_runlen, (int64_t)nruns,
                       d_counts);
    tb = 0;
    HB_TRY(rocprim::exclusive_scan(nullptr, tb, d_counts, d_starts, (int64_t)0, nruns, rocprim::plus<int64_t>(), stream));
    HB_TRY(tmp.alloc((char**)&d_tmp, tb));
    HB_TRY(rocprim::exclusive_scan(d_tmp, tb, d_counts, d_starts, (int64_t)0, nruns, rocprim::plus<int64_t>(), stream));
    HB_TRY(hipStreamSynchronize(stream));
    HB_TRY(hipGetLastError());

    tmp.release(d_keys);
    tmp.release(d_counts);
    tmp.release(d_starts);
    tmp.release(d_faces);
    out->keys = d_keys;
    out->counts = d_counts;
    out->starts = d_starts;
    out->faces = d_faces;
    out->nkeys = (int64_t)nruns;
    out->nentries = total;
    out->bitwidth = bitwidth;
    for (int k = 0; k < 6; k++) out->bbox[k] = bb.v[k];
    return hipSuccess;
}

}  // namespace pk
