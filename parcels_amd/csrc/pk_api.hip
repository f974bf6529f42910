// pk_api.hip -- host side of libparcels_hip.so: the C ABI of include/parcels_hip.h.
//
// Owns device memory (grids, field-level rings, particle columns), two HIP streams (compute + copy, so that
// the upload of the next time level overlaps the RK sub-steps of the current one) and the launch logic.
// gfx950 only; no CUDA/other-backend paths.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "pk_hashbuild.h"
#include "pk_host_stage.h"
#include "pk_kernels.h"

using namespace pk;

namespace {
std::string g_init_error;
}

struct HostGrid {
    pk_grid_desc desc;
    DGrid d;
    std::vector<void*> allocs;
    int64_t h_nentries = 0;
};

struct HostField {
    pk_field_desc desc;
    DField d;
    void* dev_data = nullptr;
    double* dev_time = nullptr;
    size_t level_bytes = 0;
    std::vector<double> time;
    bool owns_data = true;   // false for followers of a packed group (the leader owns the interleaved buffer)
    int pack_used = 1;       // leader: components handed out so far
    std::vector<int32_t> slot_level;    // committed (usable) level per ring slot, -1 = empty
    std::vector<int32_t> slot_pending;  // level being copied into the slot (async upload), -1 = none
    std::vector<uint64_t> slot_gen;     // stamp of the last upload into the slot (pk_ctx::upload_counter): tells a re-upload of the same level apart
};

constexpr int PK_STAGE_BUFFERS = 3;  // pinned staging chunks of the level stream (see stage_acquire)

// device scratch of the write filter (pk_select.inc): flag per host row, exclusive scan of the flags
struct PkSelect {
    uint32_t* d_flag = nullptr;
    uint32_t* d_offs = nullptr;
    int64_t cap = 0;
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0;
};
struct PkComm;
struct pk_ctx {
    int device = 0;
    hipStream_t compute = nullptr, copy = nullptr;
    hipStream_t probe = nullptr;     // the clock probe runs beside the advection kernel (created on first use)
    hipEvent_t ev_probe = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr;  // around the pair-copy packing of a launch (ensure_velocity_pairs)
    int fl_packs = 0;                             // pairs packed ahead of the launch in flight
    bool copy_pending = false;
    std::string err;
    std::vector<HostGrid> grids;
    std::vector<HostField> fields;
    // particles
    pk_particles_desc host{};
    DParticles dev{};
    int64_t capacity = 0;
    bool bound = false;
    // sorted-order bookkeeping
    int64_t* d_perm = nullptr;      // device row -> host row (valid when has_perm)
    bool has_perm = false;
    int64_t* d_perm_alt = nullptr;
    DParticles alt{};               // second set of columns (ping-pong target of the cell sort, staging of d2h)
    unsigned long long *d_keys = nullptr, *d_keys_alt = nullptr;
    uint32_t *d_idx = nullptr, *d_idx_alt = nullptr;
    void* d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    void* d_pack_tmp = nullptr;  // device staging of one field level before it is interleaved into a packed group
    size_t pack_tmp_bytes = 0;
    // scratch
    DCounters* d_counters = nullptr;
    // device copy of the grid / field descriptors the kernels read through KArgs::grids / fields (pk_device.h), and its pinned source
    char *d_desc = nullptr, *h_desc = nullptr;
    size_t desc_bytes = 0;
    unsigned long long* d_summary = nullptr;  // PK_NUM_STATE_CODES counts + 2 ordered-double slots
    // pinned staging ring for async level uploads
    void* stage[PK_STAGE_BUFFERS] = {};
    size_t stage_bytes[PK_STAGE_BUFFERS] = {};
    size_t stage_chunk = 0;  // bytes per staging chunk: the largest streamed level (x its packed components), at most PK_STAGE_CHUNK_BYTES
    hipEvent_t stage_ev[PK_STAGE_BUFFERS] = {};
    int stage_next = 0;
    double stage_fill_s = 0, stage_wait_s = 0, stage_bytes_total = 0;  // pk_upload_stats
    hipDeviceProp_t prop;
    // an advection launch in flight (pk_execute_begin .. pk_execute_end)
    bool in_flight = false;
    int fl_launches = 0;
    int fl_program = 0;
    // the last advection launch wrote into the second column set, which now is `dev`; `alt` still holds what it read
    bool rerun_valid = false;
    pk_exec_params rerun_prm{};
    // launcher of the run-time compiled kernel-list interpreter that carries the user kernels (pk_set_user_program)
    void (*user_launch)(const void*, int32_t, int32_t, int32_t, uint64_t, void*) = nullptr;
    int32_t user_flags = 0;
    int32_t user_nsample = 0;       // scalar fields the module's user kernels sample (PK_USER_RIDE modules)
    int32_t user_sample_fid[4] = {0, 0, 0, 0};
    // pk_particles_checkpoint: one packed device copy of every column (+ the row permutation of the cell sort)
    char* d_chk = nullptr;
    size_t chk_bytes = 0;
    bool chk_valid = false, chk_has_perm = false;
    int64_t chk_n = 0;
    bool fl_sorted = false;
    int64_t fl_n = 0;
    DCounters* h_counters = nullptr;          // pinned: the async D2H of the counters must not block the host
    DCounters h_counters0;                    // initial values of a launch's counters (pageable: the H2D copy stages it at once)
    unsigned long long* d_twe = nullptr;      // PK_MAX_TWE keys: the failing samples a launch knows (pk_exec_params.twe_key)
    unsigned long long* d_twe_found = nullptr;  // TWE_FOUND_SLOTS: hash set of the unlisted failing samples of a launch (pk_device.h: twe_note_all)
    unsigned int* d_twe_hit = nullptr;          // PK_MAX_TWE: listed sample k was justified by a lane of the launch (twe_justify)
    std::vector<int64_t> twe_found_host;        // ... of the last completed launch, ascending (pk_execute_twe_report)
    std::vector<uint8_t> twe_hit_host;
    bool twe_overflow_host = false;
    int fl_twe_n = 0;                           // listed keys of the launch in flight
    std::vector<int64_t> rerun_keys;          // the keys of the last launch (pk_execute_rerun repeats it with them)
    unsigned long long* h_summary = nullptr;  // pinned
    unsigned long long* d_tstats = nullptr;   // pk_particles_t_stats: {ordered min, ordered max, NaN count} (its own 24 bytes: not the clock-probe buffer)
    unsigned long long* d_clk = nullptr;      // clock probes around the advection kernel: [before | after][XCD 0..7]{shader-clock counter, 100 MHz counter}
    unsigned long long* h_clk = nullptr;      // pinned
    int sort_horizontal_major = -1;  // tuning knobs (environment: PK_SORT_HORIZONTAL = 0/1 forces, PK_NO_SPECIAL, PK_NO_CELL_CACHE)
    int no_special = 0;
    bool eval_points_f32 = false;  // pk_eval: the sample points are float32 particle columns (np.cos(np.deg2rad(y)) is then a float32 cosine)
    int no_cell_cache = 0;
    int no_hash_dir = 0;
    int no_cell_table = 0;
    int no_fast = 0;
    int no_fast_cgrid = 0;
    // asynchronous write-out snapshots (pk_particles_snapshot_begin / _wait): two sets of device staging columns (host row order)
    // + pinned host columns, so that the D2H and the encode of interval k overlap the launch of interval k+1
    struct Snapshot {
        void* dev[12 + PK_MAX_EXTRA] = {};
        void* host[12 + PK_MAX_EXTRA] = {};
        int64_t capacity = 0;  // rows the buffers were sized for
        int64_t n = 0;         // rows of the snapshot in flight
        uint32_t mask = 0;
        int ngrids = 0;
        size_t ss = 0;
        size_t extra_elem[PK_MAX_EXTRA] = {};
        hipEvent_t ready = nullptr, done = nullptr;
        bool in_flight = false;
    } snap[2];
    // {a, 1/width} coordinate tables of the fast A-grid path (pk_fast_agrid.h), cached per (main grid, main field)
    double* d_fast_tab = nullptr;
    size_t fast_tab_cap = 0;
    int fast_tab_grid = -1, fast_tab_field = -1;
    bool fast_tab_ok = false;
    int32_t fast_tab_off[5] = {0, 0, 0, 0, 0};
    // fast C-grid path (pk_fast_cgrid.h): per-cell records of one grid + {a, 1/width} tables of time | depth, cached per (grid, field)
    double* d_ct2 = nullptr;
    int ct2_grid = -1;
    int ct2_near = 0;  // FastC::near_edges of that grid
    // cell-packed pair copies of the staggered velocity for the 2-D dedicated C-grid kernels (pk_device.h: FastC::vp)
    char* d_vp = nullptr;
    size_t vp_bytes = 0;
    int vp_fU = -1, vp_fV = -1;
    std::vector<int32_t> vp_level;        // pair held by every pair slot, -1 = none
    std::vector<uint64_t> vp_gen;         // 4 stamps per pair slot: U and V uploads of both levels it was packed from
    int clock_probe = 0;        // measure the shader clock behind every advection kernel (option "clock_probe"; bench.py switches it on)
    bool fl_clock_probe = false;
    int no_velocity_pairs = 1;  // OPT-IN since round 5 (option "velocity_pairs" / PK_VELOCITY_PAIRS=1): packing a pair costs more than the
                                // launch it serves saves at BASELINE config 5 (pk_exec_stats.pack_ms)
    uint64_t upload_counter = 0;
    double* d_cg_tab = nullptr;
    size_t cg_tab_cap = 0;
    int cg_tab_grid = -1, cg_tab_field = -1;
    bool cg_tab_ok = false;
    int32_t cg_tab_off[3] = {0, 0, 0};

    PkSelect sel;                   // write filter on the device rows (pk_select.inc): the filtered snapshot and the multi-GPU exchange use it
    struct PkComm* comm = nullptr;  // the multi-GPU exchange (pk_comm.inc: RCCL communicator + staging), NULL until pk_comm_init

    int32_t fail(const char* where, hipError_t e) {
        err = std::string(where) + ": " + hipGetErrorString(e);
        return -1;
    }
    int32_t fail(const std::string& msg) {
        err = msg;
        return -2;
    }
};

#define PK_HIP(ctx, call)                                  \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) return (ctx)->fail(#call, e_); \
    } while (0)

// ---- small device kernels owned by this TU ------------------------------------------------------------
namespace pk {

PK_DEV unsigned long long order_double(double v) {  // order-preserving map double -> u64
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

// histogram of `state` + min/max of t over particles still in Evaluate
// Clock probe, running NEXT TO the advection kernel (its own stream): 16 single-wavefront workgroups (dealt round-robin over the 8 XCDs) each
// spin for `ticks` of the constant 100 MHz counter (s_memrealtime) and count the shader-clock cycles (s_memtime) that passed meanwhile --
// both read by the SAME wavefront, so no pairing of counters across XCDs or launches is involved (the first version of this round paired
// two probes around the kernel by XCC id and produced 2.0 ... 6.0 GHz for one and the same launch; a probe BEHIND the kernel reads the idle
// boost clock, 2.41-2.44 GHz, not the clock under the fp64 load).  One wavefront on 16 of the 1024 SIMDs: below the noise of the timed kernel.
__global__ void __launch_bounds__(64) clock_probe_kernel(unsigned long long* out, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned long long r = r0;
    while (r - r0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        r = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    out[blockIdx.x * 2] = c1 - c0;
    out[blockIdx.x * 2 + 1] = r - r0;
}

__global__ void __launch_bounds__(256) summarize_kernel(const int32_t* state, const double* t, int64_t n,
                                                        unsigned long long* out) {
    __shared__ unsigned int hist[PK_NUM_STATE_CODES];
    __shared__ unsigned long long smin, smax;
    for (int k = threadIdx.x; k < PK_NUM_STATE_CODES; k += 256) hist[k] = 0;
    if (threadIdx.x == 0) { smin = ~0ull; smax = 0ull; }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int s = state[i];
        if (s >= 0 && s < PK_NUM_STATE_CODES) atomicAdd(&hist[s], 1u);
        if (s == PK_EVALUATE) {
            unsigned long long o = order_double(t[i]);
            atomicMin(&smin, o);
            atomicMax(&smax, o);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < PK_NUM_STATE_CODES; k += 256)
        if (hist[k]) atomicAdd(&out[k], (unsigned long long)hist[k]);
    if (threadIdx.x == 0) {
        atomicMin(&out[PK_NUM_STATE_CODES], smin);
        atomicMax(&out[PK_NUM_STATE_CODES + 1], smax);
    }
}

// Field.eval / VectorField.eval at explicit points (no particles): what >= 0 scalar field, -1 UV, -2 UVW
template <class FT, int INTERP, bool TYPED>
__global__ void __launch_bounds__(256) eval_kernel(const KArgs a, int what, int64_t m, const double* t, const double* z,
                                                   const double* y, const double* x, double* ou, double* ov, double* ow,
                                                   int32_t* ost) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const DField& mf = kfield(a, a.main_field);
    const DGrid& mg = kgrid(a, a.main_grid);
    Coords mc{CellCache{nullptr, nullptr, nullptr}, mf.time, mg.depth, mg.lat, mg.lon, mf.tfirst, mf.tlast, mg.zfirst, mg.zlast, mg.yfirst, mg.ylast, mg.xfirst, mg.xlast};
    PCtx c;
    c.state = PK_EVALUATE;
    c.pf = false;
    c.hz = c.hy = c.hx = c.ht = 0;
    c.hyx_valid = false;
    c.first_eval = 0xFu;
    c.u32 = c.v32 = false;
    c.oob = false;
    c.ei0 = c.ei1 = c.ei2 = c.ei3 = 0;
    c.it = 0u;  // not inside the loop of kernel.py:190: the caller of pk_eval owns the batch (parcels_amd/field.py: _eval_key)
    c.klo = 0;
    const bool pos_f32 = a.prm.reset_state != 0;  // (pk_eval's own use of the field: option "eval_points_f32")
    c.zpos_f32 = pos_f32;
    if (what < 0) {
        double u, v, w;
        eval_uvw<FT, -1, INTERP, TYPED>(a, mc, c, what == -2, t[i], z[i], y[i], x[i], pos_f32, u, v, w);
        ou[i] = u;
        if (ov) ov[i] = v;
        if (ow) ow[i] = w;
    } else {
        ou[i] = eval_scalar<FT, TYPED>(a, mc, c, what, t[i], z[i], y[i], x[i], pos_f32);
    }
    if (ost) ost[i] = c.state | (c.oob ? PK_EVAL_MASKED : 0);
}

// One pk_eval call is one batch of the reference: lenT = 2 if np.any(tau > 0) else 1, lenZ likewise over the WHOLE batch
// (_xinterpolators.py:130-131,401-402,575-576).  flags: bit 0 any(tau > 0), bit 1 any(zeta > 0).
__global__ void __launch_bounds__(256) batch_len_kernel(const DField f, const DGrid g, int64_t m, const double* t, const double* z, unsigned* flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned fl = 0;
    if (i < m) {
        GPos p;
        if (time_search(f, f.time, t[i], 0, p) && p.tau > 0) fl |= 1u;
        if (g.has_z) {
            search_1d(g.depth, g.nz, g.depth[0], g.depth[g.nz - 1], z[i], g.depth_f32 != 0, false, 0, p.zi, p.zeta);
            if (p.zeta > 0) fl |= 2u;
        }
    }
    if (__any(fl & 1u) && (threadIdx.x & 63) == 0) atomicOr(flags, 1u);
    if (__any(fl & 2u) && (threadIdx.x & 63) == 0) atomicOr(flags, 2u);
}

// XGrid.search + ravel_index without a guess (ParticleSet.populate_indices, particleset.py:252-262)
__global__ void __launch_bounds__(256) search_kernel(const DGrid g, int64_t m, const double* z, const double* y, const double* x,
                                                     int32_t* ei_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    PCtx c;
    c.state = PK_EVALUATE;
    c.pf = false;
    c.hz = c.hy = c.hx = c.ht = 0;
    c.hyx_valid = false;
    c.zpos_f32 = false;
    GPos p;
    int32_t ei = 0;
    grid_search<-1, false>(g, nullptr, z[i], y[i], x[i], false, &ei, c, false, p);  // indices only: dtype emulation of the bcoords is irrelevant
    ei_out[i] = ei;
}

// ---- cell sort ----------------------------------------------------------------------------------------
// key = linear cell index of the particle's current position on the main grid (z-major like the field layout),
// so that the 64 lanes of a wavefront gather from neighbouring cells (coalesced, L2-friendly).
__global__ void __launch_bounds__(256) sort_key_kernel(const DGrid g, const DParticles P, unsigned long long* keys, uint32_t* idx,
                                                       int horizontal_major) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n) return;
    const bool pf = P.spatial_f32 != 0;
    const double z = pf ? (double)((const float*)P.z)[i] : ((const double*)P.z)[i];
    const double y = pf ? (double)((const float*)P.y)[i] : ((const double*)P.y)[i];
    const double x = pf ? (double)((const float*)P.x)[i] : ((const double*)P.x)[i];
    unsigned long long zi = 0, h = 0;
    if (g.has_z && g.nz >= 2) zi = (unsigned long long)cell_index(g.depth, g.nz, z, 0);
    if (g.kind == 1) {
        // depth-major like the field layout, then the 30-bit Morton code of the hash grid (neighbouring codes = neighbouring
        // cells).  Measured on C3: depth-major 295 ms vs horizontal-major 308 ms per 3.6e8 particle-steps.
        h = morton_code(g, make_qpoint(g, y, x));
        keys[i] = horizontal_major ? ((h << 12) | (zi & 0xFFFull)) : ((zi << 30) | h);
    } else {
        unsigned long long yi = (g.has_y && g.ny >= 2) ? (unsigned long long)cell_index(g.lat, g.ny, y, 0) : 0;
        unsigned long long xi = (g.has_x && g.nx >= 2) ? (unsigned long long)cell_index(g.lon, g.nx, x, 0) : 0;
        keys[i] = (zi * (unsigned long long)g.ny + yi) * (unsigned long long)g.nx + xi;
    }
    idx[i] = (uint32_t)i;
}

template <class T>
__global__ void __launch_bounds__(256) gather_rows_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                          const uint32_t* __restrict__ perm, int64_t n, int width) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t src = perm[i];
    for (int k = 0; k < width; k++) out[i * width + k] = in[src * width + k];
}
// out[i] = in[perm[i]] with the int64 device-row -> host-row map of the cell sort (pk_particles_h2d_columns)
template <class T>
__global__ void __launch_bounds__(256) gather_rows64_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                            const int64_t* __restrict__ perm, int64_t n, int width) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t src = perm[i];
    for (int k = 0; k < width; k++) out[i * width + k] = in[src * width + k];
}
__global__ void __launch_bounds__(256) fill_f64_kernel(double* __restrict__ out, int64_t n, double v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v;
}
// smallest / largest finite-or-infinite (non-NaN) t as order-preserving integers, and the NaN count: out[0] = min, out[1] = max, out[2] = NaNs
__global__ void __launch_bounds__(256) t_stats_kernel(const double* __restrict__ t, int64_t n, unsigned long long* __restrict__ out) {
    unsigned long long lo = ~0ull, hi = 0ull, nn = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = t[i];
        if (v != v) { nn++; continue; }
        unsigned long long b;
        memcpy(&b, &v, 8);
        b = (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);  // order-preserving map of the doubles
        lo = b < lo ? b : lo;
        hi = b > hi ? b : hi;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o), n2 = __shfl_xor(nn, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
        nn += n2;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&out[0], lo);
        atomicMax(&out[1], hi);
        if (nn) atomicAdd(&out[2], nn);
    }
}
template <class T>
__global__ void __launch_bounds__(256) scatter_rows_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                           const int64_t* __restrict__ perm, int64_t n, int width) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t dst = perm[i];
    for (int k = 0; k < width; k++) out[dst * width + k] = in[i * width + k];
}
__global__ void __launch_bounds__(256) compose_perm_kernel(const int64_t* __restrict__ old_perm, const uint32_t* __restrict__ perm,
                                                           int64_t* __restrict__ new_perm, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    new_perm[i] = old_perm ? old_perm[perm[i]] : (int64_t)perm[i];
}

// ---- device-side removal of deleted particles (Kernel.remove_deleted, kernel.py:98-106) ----------------------------------
__global__ void __launch_bounds__(256) keep_flags_kernel(const int32_t* __restrict__ state, const int64_t* __restrict__ perm, int64_t n,
                                                         unsigned long long* __restrict__ keep_dev, uint32_t* __restrict__ keep_host) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned k = state[i] != PK_DELETE;
    keep_dev[i] = k;
    keep_host[perm ? perm[i] : i] = k;
}
template <class T>
__global__ void __launch_bounds__(256) compact_rows_kernel(const T* __restrict__ in, T* __restrict__ out, const unsigned long long* __restrict__ keep,
                                                           const unsigned long long* __restrict__ pos, int64_t n, int width) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const int64_t dst = (int64_t)pos[i];
    for (int k = 0; k < width; k++) out[dst * width + k] = in[i * width + k];
}
// new_perm[new device row] = new host row of the old host row this device row mapped to
__global__ void __launch_bounds__(256) compact_perm_kernel(const int64_t* __restrict__ perm, const unsigned long long* __restrict__ keep,
                                                           const unsigned long long* __restrict__ pos, const uint32_t* __restrict__ host_pos,
                                                           int64_t n, int64_t* __restrict__ new_perm) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !keep[i]) return;
    new_perm[pos[i]] = (int64_t)host_pos[perm[i]];
}

// scatter one contiguous field level into its component slot of a packed (array-of-structs) group buffer
// Cell-packed pair copy of the staggered velocity (FastC::vp): for every element e = (zi, yi, xi) of a level the four values the 2-D
// C-grid kernels read for cell e -- U at e + dU0 / dU1, V at e + dV0 / dV1 (byte offsets into the level rings, pk_fast_cgrid.h) -- of
// level slot `o0`, then of level slot `o1`.  Elements in the last row / column are no cells (their offsets leave the level): zeros.
template <class T>
__global__ void __launch_bounds__(256) cg_pack_pairs_kernel(const char* __restrict__ U, const char* __restrict__ V, int64_t dU0, int64_t dU1,
                                                            int64_t dV0, int64_t dV1, int64_t cb, int64_t o0, int64_t o1, int64_t n, int32_t ny,
                                                            int32_t nx, T* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int xi = (int)(e % nx), yi = (int)((e / nx) % ny);
        T v[8];
        if (xi < nx - 1 && yi < ny - 1) {
            const int64_t b0 = e * cb + o0, b1 = e * cb + o1;
            v[0] = *reinterpret_cast<const T*>(U + dU0 + b0);
            v[1] = *reinterpret_cast<const T*>(U + dU1 + b0);
            v[2] = *reinterpret_cast<const T*>(V + dV0 + b0);
            v[3] = *reinterpret_cast<const T*>(V + dV1 + b0);
            v[4] = *reinterpret_cast<const T*>(U + dU0 + b1);
            v[5] = *reinterpret_cast<const T*>(U + dU1 + b1);
            v[6] = *reinterpret_cast<const T*>(V + dV0 + b1);
            v[7] = *reinterpret_cast<const T*>(V + dV1 + b1);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = (T)0;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) out[e * 8 + k] = v[k];
    }
}

template <class T>
__global__ void __launch_bounds__(256) interleave_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t n, int ncomp) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i * ncomp] = src[i];
}

// device-to-device float4 copy used as the measured HBM ceiling: 4 independent 16-byte loads in flight per lane, one block
// per 4 KiB-per-wave tile (no grid-stride tail), which is what reaches the ~6.3 TB/s of MI355X_MICROARCH.md
__global__ void __launch_bounds__(256) copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (base + 768 < n) {
        const float4 a = src[base], b = src[base + 256], c = src[base + 512], d = src[base + 768];
        dst[base] = a; dst[base + 256] = b; dst[base + 512] = c; dst[base + 768] = d;
    } else {
        for (int k = 0; k < 4; k++)
            if (base + 256 * k < n) dst[base + 256 * k] = src[base + 256 * k];
    }
}

}  // namespace pk

// pageable -> pinned staging: a ring of PK_STAGE_BUFFERS fixed-size pinned chunks (allocated once; pinning a whole 4 GB level costs
// ~1 s).  The host fills chunk k+1 (pk_host_stage.cpp: persistent thread pool, AVX2 interleave, non-temporal stores) while chunk k
// is on the wire; a third chunk keeps the DMA queue from draining while the fill of the next one is being handed to the pool.
constexpr size_t PK_STAGE_CHUNK_BYTES = (size_t)256 << 20;
using pkhost::parallel_interleave;
using pkhost::parallel_memcpy;
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// the next staging chunk, free of its previous DMA (blocks on that DMA's event; the wait is accounted in stage_wait_s)
static int32_t stage_acquire(pk_ctx* ctx, int* out) {
    const int k = ctx->stage_next;
    ctx->stage_next = (k + 1) % PK_STAGE_BUFFERS;
    if (!ctx->stage[k]) {
        if (!ctx->stage_chunk) ctx->stage_chunk = PK_STAGE_CHUNK_BYTES;
        PK_HIP(ctx, hipHostMalloc(&ctx->stage[k], ctx->stage_chunk, hipHostMallocDefault));
        ctx->stage_bytes[k] = ctx->stage_chunk;
    } else {
        const double t0 = now_s();
        PK_HIP(ctx, hipEventSynchronize(ctx->stage_ev[k]));
        ctx->stage_wait_s += now_s() - t0;
    }
    *out = k;
    return 0;
}
static int32_t stage_submit(pk_ctx* ctx, int k, void* dst, size_t bytes) {
    PK_HIP(ctx, hipMemcpyAsync(dst, ctx->stage[k], bytes, hipMemcpyHostToDevice, ctx->copy));
    PK_HIP(ctx, hipEventRecord(ctx->stage_ev[k], ctx->copy));
    ctx->stage_bytes_total += (double)bytes;
    return 0;
}

// DGrid::cell_tab: one record per cell, written by the device functions the search would otherwise run on every cache miss
// (identical instruction sequences, so the values are the ones the per-miss computation produces)
__global__ void cell_table_kernel(const pk::DGrid g, double* tab) {
    using namespace pk;
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= (int64_t)(g.ny - 1) * g.nx) return;
    const int xi = (int)(cell % g.nx);
    double* out = tab + cell * CT_STRIDE;
    if (xi >= g.nx - 1) {  // not a cell (last node of a row): never read
        for (int k = 0; k < CT_STRIDE; k++) out[k] = 0.0;
        return;
    }
    const double* r0 = g.node_tab + cell * 5;
    const double* r1 = r0 + (int64_t)g.nx * 5;
    // corner order c0=(yi,xi) c1=(yi,xi+1) c2=(yi+1,xi+1) c3=(yi+1,xi)
    const double* nodes[4] = {r0, r0 + 5, r1 + 5, r1};
    double lon[4], lat[4], cX[4], cY[4], cZ[4];
    for (int k = 0; k < 4; k++) { lon[k] = nodes[k][0]; lat[k] = nodes[k][1]; cX[k] = nodes[k][2]; cY[k] = nodes[k][3]; cZ[k] = nodes[k][4]; }
    double eu[3] = {0, 0, 0}, ev[3] = {0, 0, 0}, pu[4] = {0, 0, 0, 0}, pv[4] = {0, 0, 0, 0};
    unsigned long long box;
    if (g.spherical) {
        spherical_project_cell(cX, cY, cZ, eu, ev, pu, pv);
        box = pack_quantised_box(g, cX, cY, cZ, true);
    } else {
        box = pack_quantised_box(g, lon, lat, lat, false);
    }
    for (int k = 0; k < 4; k++) { out[2 * k] = lon[k]; out[2 * k + 1] = lat[k]; }
    for (int k = 0; k < 3; k++) { out[8 + k] = eu[k]; out[11 + k] = ev[k]; }
    for (int k = 0; k < 4; k++) { out[14 + k] = pu[k]; out[18 + k] = pv[k]; }
    out[22] = __longlong_as_double((long long)box);
    out[23] = 0.0;
}

// FastC::ct2 (pk_fast_cgrid.h): per cell, the record of cell_table_kernel re-expressed for the dedicated C-grid kernels -- the
// query-independent sub-expressions of bilinear_inverse in ITS evaluation order (so that finishing them with the query point gives
// the bits bilinear_inverse gives) and the corner longitudes after CGrid_Velocity's antimeridian unwrapping (cgrid_velocity).
__global__ void cell_table2_kernel(const pk::DGrid g, double* tab, int* wide) {
    using namespace pk;
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= (int64_t)(g.ny - 1) * g.nx) return;
    const int xi = (int)(cell % g.nx);
    double* out = tab + cell * CT2_STRIDE;
    for (int k = 0; k < CT2_STRIDE; k++) out[k] = 0.0;
    if (xi >= g.nx - 1) return;  // not a cell (last node of a row): never read
    const double* ct = g.cell_tab + cell * CT_STRIDE;
    double px[4], py[4];
    for (int k = 0; k < 4; k++) { px[k] = ct[14 + k]; py[k] = ct[18 + k]; }
    const double a0 = px[0];
    const double a1 = -px[0] + px[1];
    const double a2 = -px[0] + px[3];
    const double a3 = ((px[0] + -px[1]) + px[2]) + -px[3];
    const double b0 = py[0];
    const double b1 = -py[0] + py[1];
    const double b2 = -py[0] + py[3];
    const double b3 = ((py[0] + -py[1]) + py[2]) + -py[3];
    const double aa = a3 * b2 - a2 * b3;
    const double bb0 = a3 * b0 - a0 * b3 + a1 * b2 - a2 * b1;  // bb = bb0 + xq * b3 - yq * a3
    const double cc0 = a1 * b0 - a0 * b1;                      // cc = cc0 + xq * b1 - yq * a1
    for (int k = 0; k < 6; k++) out[k] = ct[8 + k];  // eu, ev
    out[6] = a0; out[7] = a1; out[8] = a2; out[9] = a3; out[10] = b1; out[11] = b3;
    out[12] = 4 * aa;
    out[13] = bb0;
    out[14] = cc0;
    out[15] = ct[22];  // quantised hash box
    double lon[4];
    for (int k = 0; k < 4; k++) lon[k] = pymod360(ct[2 * k] + 180.0) - 180.0;  // _xinterpolators.py:230-233
    for (int k = 1; k < 4; k++)
        if (lon[k] - lon[0] > 180) lon[k] = lon[k] - 360;
    for (int k = 1; k < 4; k++)
        if (-lon[k] + lon[0] > 180) lon[k] = lon[k] + 360;
    for (int k = 0; k < 4; k++) { out[16 + k] = lon[k]; out[20 + k] = ct[2 * k + 1]; out[24 + k] = py[k]; }
    // FastC::near_edges (pk_fast_cgrid.h, cos_near): every cell spans less than 2^-8 rad of latitude, so the edge points of CGrid_Velocity's
    // geodetic distances lie within 2^-7 rad of any sample point found in the cell
    const double la_lo = fmin(fmin(ct[1], ct[3]), fmin(ct[5], ct[7])), la_hi = fmax(fmax(ct[1], ct[3]), fmax(ct[5], ct[7]));
    if (!((la_hi - la_lo) * DEG2RAD <= 0.00390625)) atomicOr(wide, 1);
}

// Cells on which the reference's bilinear inverse (index_search.py:122-177, restated in bilinear_inverse) is numerically unreliable:
// the quadratic branch is taken (|aa| >= 1e-12) although the cell is a parallelogram to rounding, so (-bb + sqrt(bb^2 - 4 aa cc)) /
// (2 aa) cancels completely and the point-in-cell test of such a cell accepts points that lie elsewhere.  (A flat mesh in metres has
// aa ~ 1e-9 of pure rounding noise on an exact parallelogram; in degrees or on the unit sphere the same noise is below the 1e-12
// switch to the linear branch.)  The reference then returns whichever listed face comes first in table order; neighbour-first
// probing assumes at most one face contains the point, so it must be off on such a mesh.  Found by seed 2985 of the fuzz generator.
__global__ void illconditioned_cell_kernel(const pk::DGrid g, int* flag) {
    using namespace pk;
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= (int64_t)(g.ny - 1) * g.nx || (int)(cell % g.nx) >= g.nx - 1) return;
    const double* r0 = g.node_tab + cell * 5;
    const double* r1 = r0 + (int64_t)g.nx * 5;
    const double* nodes[4] = {r0, r0 + 5, r1 + 5, r1};
    double px[4], py[4];
    if (g.spherical) {
        double cX[4], cY[4], cZ[4], eu[3], ev[3];
        for (int k = 0; k < 4; k++) { cX[k] = nodes[k][2]; cY[k] = nodes[k][3]; cZ[k] = nodes[k][4]; }
        spherical_project_cell(cX, cY, cZ, eu, ev, px, py);
    } else {
        for (int k = 0; k < 4; k++) { px[k] = nodes[k][0]; py[k] = nodes[k][1]; }
    }
    const double a1 = -px[0] + px[1], a2 = -px[0] + px[3], a3 = ((px[0] + -px[1]) + px[2]) + -px[3];
    const double b1 = -py[0] + py[1], b2 = -py[0] + py[3], b3 = ((py[0] + -py[1]) + py[2]) + -py[3];
    const double aa = a3 * b2 - a2 * b3;
    const double scale = fmax(fmax(fabs(a1 * b2), fabs(a2 * b1)), fmax(fabs(a3 * b2), fabs(a2 * b3)));
    if (fabs(aa) >= 1e-12 && fabs(aa) < 1e-9 * scale) atomicOr(flag, 1);
}

template <class T>
static int32_t upload(pk_ctx* ctx, HostGrid& g, const T* host, size_t n, const T** dev) {
    *dev = nullptr;
    if (!host || n == 0) return 0;
    void* p = nullptr;
    PK_HIP(ctx, hipMalloc(&p, n * sizeof(T)));
    g.allocs.push_back(p);
    PK_HIP(ctx, hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    *dev = (const T*)p;
    return 0;
}


// ---- context -------------------------------------------------------------------------------------------
extern "C" {

int32_t pk_abi_version(void) { return PK_ABI_VERSION; }

const char* pk_last_error(const pk_ctx* ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int32_t pk_init(int32_t device, pk_ctx** out) {
    if (!out) return -2;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        g_init_error = std::string("no HIP device available: ") + hipGetErrorString(e);
        return -1;
    }
    if (device < 0 || device >= ndev) {
        g_init_error = "device index out of range";
        return -2;
    }
    pk_ctx* ctx = new pk_ctx();
    ctx->device = device;
    if (const char* e = getenv("PK_SORT_HORIZONTAL")) ctx->sort_horizontal_major = atoi(e);
    if (const char* e = getenv("PK_NO_SPECIAL")) ctx->no_special = atoi(e);
    if (const char* e = getenv("PK_NO_CELL_CACHE")) ctx->no_cell_cache = atoi(e);
    if (const char* e = getenv("PK_NO_HASH_DIR")) ctx->no_hash_dir = atoi(e);
    if (const char* e = getenv("PK_NO_CELL_TABLE")) ctx->no_cell_table = atoi(e);
    if (const char* e = getenv("PK_NO_FAST")) ctx->no_fast = atoi(e);
    if (const char* e = getenv("PK_NO_FAST_CGRID")) ctx->no_fast_cgrid = atoi(e);
    if (const char* e = getenv("PK_NO_VELOCITY_PAIRS")) ctx->no_velocity_pairs = atoi(e);
    if (const char* e = getenv("PK_VELOCITY_PAIRS")) ctx->no_velocity_pairs = !atoi(e);
    *out = ctx;
    PK_HIP(ctx, hipSetDevice(device));
    PK_HIP(ctx, hipGetDeviceProperties(&ctx->prop, device));
    PK_HIP(ctx, hipStreamCreateWithFlags(&ctx->compute, hipStreamNonBlocking));
    PK_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking));
    PK_HIP(ctx, hipEventCreate(&ctx->ev0));
    PK_HIP(ctx, hipEventCreate(&ctx->ev1));
    PK_HIP(ctx, hipEventCreate(&ctx->ev2));
    PK_HIP(ctx, hipEventCreate(&ctx->ev_p0));
    PK_HIP(ctx, hipEventCreate(&ctx->ev_p1));
    for (int k = 0; k < PK_STAGE_BUFFERS; k++) PK_HIP(ctx, hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
    PK_HIP(ctx, hipMalloc((void**)&ctx->d_counters, sizeof(DCounters)));
    PK_HIP(ctx, hipMalloc((void**)&ctx->d_twe, sizeof(unsigned long long) * PK_MAX_TWE));
    PK_HIP(ctx, hipMalloc((void**)&ctx->d_twe_found, sizeof(unsigned long long) * TWE_FOUND_SLOTS));
    PK_HIP(ctx, hipMalloc((void**)&ctx->d_twe_hit, sizeof(unsigned int) * PK_MAX_TWE));
    PK_HIP(ctx, hipMalloc((void**)&ctx->d_summary, sizeof(unsigned long long) * (PK_NUM_STATE_CODES + 2)));
    PK_HIP(ctx, hipHostMalloc((void**)&ctx->h_counters, sizeof(DCounters), hipHostMallocDefault));
    PK_HIP(ctx, hipHostMalloc((void**)&ctx->h_summary, sizeof(unsigned long long) * (PK_NUM_STATE_CODES + 2), hipHostMallocDefault));
    PK_HIP(ctx, hipMalloc((void**)&ctx->d_clk, sizeof(unsigned long long) * 32));
    PK_HIP(ctx, hipMalloc((void**)&ctx->d_tstats, sizeof(unsigned long long) * 3));
    PK_HIP(ctx, hipHostMalloc((void**)&ctx->h_clk, sizeof(unsigned long long) * 32, hipHostMallocDefault));
    return 0;
}

int32_t pk_set_option(pk_ctx* ctx, const char* name, int32_t value) {
    if (!ctx || !name) return -2;
    const std::string n(name);
    if (n == "fast_path") ctx->no_fast = !value;
    else if (n == "fast_cgrid") ctx->no_fast_cgrid = !value;
    else if (n == "velocity_pairs") ctx->no_velocity_pairs = !value;
    else if (n == "clock_probe") ctx->clock_probe = value;
    else if (n == "special_programs") ctx->no_special = !value;
    else if (n == "cell_cache") ctx->no_cell_cache = !value;
    else if (n == "hash_directory") ctx->no_hash_dir = !value;
    else if (n == "cell_table") ctx->no_cell_table = !value;
    else if (n == "sort_horizontal") ctx->sort_horizontal_major = value;
    else if (n == "eval_points_f32") ctx->eval_points_f32 = value != 0;
    else return ctx->fail("pk_set_option: unknown option '" + n + "'");
    return 0;
}

int32_t pk_upload_stats(pk_ctx* ctx, double* out4) {
    if (!ctx || !out4) return -2;
    out4[0] = ctx->stage_fill_s;       // host threads filling pinned chunks
    out4[1] = ctx->stage_wait_s;       // blocked on the DMA that still reads the chunk to be refilled
    out4[2] = ctx->stage_bytes_total;  // bytes handed to the DMA engine through the staging ring
    out4[3] = (double)pkhost::copy_threads();
    return 0;
}

static void free_snapshots(pk_ctx* ctx) {
    for (auto& sn : ctx->snap) {
        if (sn.in_flight && sn.done) (void)hipEventSynchronize(sn.done);
        for (int k = 0; k < 12 + PK_MAX_EXTRA; k++) {
            if (sn.dev[k]) (void)hipFree(sn.dev[k]);
            if (sn.host[k]) (void)hipHostFree(sn.host[k]);
            sn.dev[k] = sn.host[k] = nullptr;
        }
        sn.capacity = 0;
        sn.in_flight = false;
    }
}

static void free_particles(pk_ctx* ctx) {
    if (ctx->d_chk) (void)hipFree(ctx->d_chk);
    ctx->d_chk = nullptr;
    ctx->chk_bytes = 0;
    ctx->chk_valid = false;
    void* cols[] = {ctx->dev.t,  ctx->dev.z,  ctx->dev.y,       ctx->dev.x,     ctx->dev.dz, ctx->dev.dy,
                    ctx->dev.dx, ctx->dev.dt, ctx->dev.next_dt, ctx->dev.state, ctx->dev.ei, ctx->dev.particle_id, ctx->dev.iter, ctx->alt.iter};
    ctx->rerun_valid = false;
    for (void* p : cols)
        if (p) (void)hipFree(p);
    void* alts[] = {ctx->alt.t,  ctx->alt.z,  ctx->alt.y,       ctx->alt.x,     ctx->alt.dz, ctx->alt.dy,
                    ctx->alt.dx, ctx->alt.dt, ctx->alt.next_dt, ctx->alt.state, ctx->alt.ei, ctx->alt.particle_id,
                    ctx->d_perm, ctx->d_perm_alt, ctx->d_keys, ctx->d_keys_alt, ctx->d_idx, ctx->d_idx_alt, ctx->d_sort_tmp};
    for (void* p : alts)
        if (p) (void)hipFree(p);
    for (int k = 0; k < PK_MAX_EXTRA; k++) {
        if (ctx->dev.extra[k]) (void)hipFree(ctx->dev.extra[k]);
        if (ctx->alt.extra[k]) (void)hipFree(ctx->alt.extra[k]);
    }
    ctx->d_perm = ctx->d_perm_alt = nullptr;
    ctx->d_keys = ctx->d_keys_alt = nullptr;
    ctx->d_idx = ctx->d_idx_alt = nullptr;
    ctx->d_sort_tmp = nullptr;
    ctx->sort_tmp_bytes = 0;
    ctx->dev = DParticles{};
    ctx->alt = DParticles{};
    ctx->capacity = 0;
}

int32_t pk_destroy(pk_ctx* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& g : ctx->grids)
        for (void* p : g.allocs) (void)hipFree(p);
    for (auto& f : ctx->fields) {
        if (f.dev_data && f.owns_data) (void)hipFree(f.dev_data);
        if (f.dev_time) (void)hipFree(f.dev_time);
    }
    free_particles(ctx);
    free_snapshots(ctx);
    for (auto& sn : ctx->snap) {
        if (sn.ready) (void)hipEventDestroy(sn.ready);
        if (sn.done) (void)hipEventDestroy(sn.done);
    }
    if (ctx->d_pack_tmp) (void)hipFree(ctx->d_pack_tmp);
    if (ctx->d_fast_tab) (void)hipFree(ctx->d_fast_tab);
    if (ctx->d_cg_tab) (void)hipFree(ctx->d_cg_tab);
    if (ctx->d_ct2) (void)hipFree(ctx->d_ct2);
    if (ctx->d_vp) (void)hipFree(ctx->d_vp);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_twe) (void)hipFree(ctx->d_twe);
    if (ctx->d_twe_found) (void)hipFree(ctx->d_twe_found);
    if (ctx->d_twe_hit) (void)hipFree(ctx->d_twe_hit);
    if (ctx->d_desc) (void)hipFree(ctx->d_desc);
    if (ctx->h_desc) (void)hipHostFree(ctx->h_desc);
    if (ctx->d_summary) (void)hipFree(ctx->d_summary);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    if (ctx->h_summary) (void)hipHostFree(ctx->h_summary);
    (void)pk_comm_destroy(ctx);
    if (ctx->sel.d_flag) (void)hipFree(ctx->sel.d_flag);
    if (ctx->sel.d_offs) (void)hipFree(ctx->sel.d_offs);
    if (ctx->sel.d_tmp) (void)hipFree(ctx->sel.d_tmp);
    if (ctx->d_clk) (void)hipFree(ctx->d_clk);
    if (ctx->d_tstats) (void)hipFree(ctx->d_tstats);
    if (ctx->h_clk) (void)hipHostFree(ctx->h_clk);
    for (int k = 0; k < PK_STAGE_BUFFERS; k++) {
        if (ctx->stage[k]) (void)hipHostFree(ctx->stage[k]);
        if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
    }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
    if (ctx->ev_p0) (void)hipEventDestroy(ctx->ev_p0);
    if (ctx->ev_p1) (void)hipEventDestroy(ctx->ev_p1);
    if (ctx->compute) (void)hipStreamDestroy(ctx->compute);
    if (ctx->copy) (void)hipStreamDestroy(ctx->copy);
    if (ctx->probe) (void)hipStreamDestroy(ctx->probe);
    if (ctx->ev_probe) (void)hipEventDestroy(ctx->ev_probe);
    delete ctx;
    return 0;
}

int32_t pk_get_device_info(pk_ctx* ctx, pk_device_info* out) {
    if (!ctx || !out) return -2;
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", ctx->prop.name);
    snprintf(out->arch, sizeof(out->arch), "%s", ctx->prop.gcnArchName);
    out->compute_units = ctx->prop.multiProcessorCount;
    out->wavefront_size = ctx->prop.warpSize;
    out->lds_bytes_per_block = (int32_t)ctx->prop.sharedMemPerBlock;
    out->clock_khz = ctx->prop.clockRate;
    size_t fr = 0, tot = 0;
    PK_HIP(ctx, hipSetDevice(ctx->device));
    PK_HIP(ctx, hipMemGetInfo(&fr, &tot));
    out->total_mem = (int64_t)tot;
    out->free_mem = (int64_t)fr;
    return 0;
}

// ---- grids ---------------------------------------------------------------------------------------------
int32_t pk_grid_create(pk_ctx* ctx, const pk_grid_desc* desc, int32_t* grid_id) {
    if (!ctx || !desc || !grid_id) return -2;
    if ((int)ctx->grids.size() >= PK_MAX_GRIDS) return ctx->fail("too many grids (PK_MAX_GRIDS)");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    ctx->grids.emplace_back();
    HostGrid& g = ctx->grids.back();
    g.desc = *desc;
    DGrid& d = g.d;
    memset(&d, 0, sizeof(d));
    d.kind = desc->kind;
    d.spherical = desc->spherical;
    d.has_x = desc->has_x; d.has_y = desc->has_y; d.has_z = desc->has_z;
    d.nx = desc->nx; d.ny = desc->ny; d.nz = desc->nz;
    d.xdim = desc->xdim; d.ydim = desc->ydim; d.zdim = desc->zdim;
    d.off_x = desc->off_x; d.off_y = desc->off_y; d.off_z = desc->off_z;
    d.lon_f32 = desc->lon_f32; d.lat_f32 = desc->lat_f32; d.depth_f32 = desc->depth_f32;
    d.deg2m = desc->deg2m;
    if (desc->depth && desc->nz > 0) { d.zfirst = desc->depth[0]; d.zlast = desc->depth[desc->nz - 1]; }
    if (desc->kind == 0) {
        if (desc->lat && desc->ny > 0) { d.yfirst = desc->lat[0]; d.ylast = desc->lat[desc->ny - 1]; }
        if (desc->lon && desc->nx > 0) { d.xfirst = desc->lon[0]; d.xlast = desc->lon[desc->nx - 1]; }
    }
    const size_t nlon = desc->kind == 1 ? (size_t)desc->ny * desc->nx : (size_t)desc->nx;
    const size_t nlat = desc->kind == 1 ? (size_t)desc->ny * desc->nx : (size_t)desc->ny;
    int32_t rc;
    if (desc->kind == 1) {
        // array-of-structs node table {lon, lat, X, Y, Z}: one or two cache lines per cell row instead of five planes
        if (desc->spherical && !desc->node_xyz) return ctx->fail("spherical curvilinear grid needs node_xyz (unit-sphere node coordinates)");
        std::vector<double> tab(nlon * 5);
        for (size_t k = 0; k < nlon; k++) {
            tab[5 * k + 0] = desc->lon[k];
            tab[5 * k + 1] = desc->lat[k];
            tab[5 * k + 2] = desc->spherical ? desc->node_xyz[k] : 0.0;
            tab[5 * k + 3] = desc->spherical ? desc->node_xyz[nlon + k] : 0.0;
            tab[5 * k + 4] = desc->spherical ? desc->node_xyz[2 * nlon + k] : 0.0;
        }
        if ((rc = upload(ctx, g, (const double*)tab.data(), tab.size(), &d.node_tab))) return rc;
    } else {
        if ((rc = upload(ctx, g, desc->lon, nlon, &d.lon))) return rc;
        if ((rc = upload(ctx, g, desc->lat, nlat, &d.lat))) return rc;
    }
    if ((rc = upload(ctx, g, desc->depth, (size_t)desc->nz, &d.depth))) return rc;
    if (desc->kind == 1) {
        if (desc->h_keys && desc->h_nkeys > 0) {  // table built by the caller (e.g. the reference's own SpatialHash)
            if ((rc = upload(ctx, g, desc->h_keys, (size_t)desc->h_nkeys, &d.h_keys))) return rc;
            if ((rc = upload(ctx, g, desc->h_starts, (size_t)desc->h_nkeys, &d.h_starts))) return rc;
            if ((rc = upload(ctx, g, desc->h_counts, (size_t)desc->h_nkeys, &d.h_counts))) return rc;
            if ((rc = upload(ctx, g, desc->h_faces, (size_t)desc->h_nentries, &d.h_faces))) return rc;
            d.h_nkeys = desc->h_nkeys;
            g.h_nentries = desc->h_nentries;
            d.h_bitwidth = desc->h_bitwidth;
            for (int k = 0; k < 6; k++) d.h_bbox[k] = desc->h_bbox[k];
        } else {  // built here, on the device, from the node table
            HashBuildResult hb;
            std::string msg;
            hipError_t e = build_spatial_hash(ctx->compute, d.node_tab, desc->ny, desc->nx, desc->spherical, &hb, &msg);
            if (e != hipSuccess) return ctx->fail("spatial hash build: " + msg);
            g.allocs.push_back(hb.keys); g.allocs.push_back(hb.starts); g.allocs.push_back(hb.counts); g.allocs.push_back(hb.faces);
            d.h_keys = hb.keys; d.h_starts = hb.starts; d.h_counts = hb.counts; d.h_faces = hb.faces;
            d.h_nkeys = hb.nkeys;
            g.h_nentries = hb.nentries;
            d.h_bitwidth = hb.bitwidth;
            for (int k = 0; k < 6; k++) d.h_bbox[k] = hb.bbox[k];
        }
        for (int k = 0; k < 3; k++) {  // reciprocal widths of the hash grid for quantize() (pk_device.h: div_by_recip)
            const double w = d.h_bbox[2 * k + 1] - d.h_bbox[2 * k], r = 1.0 / w;
            d.h_rinv[k] = (w >= 1e-100 && w <= 1e100 && std::isfinite(r)) ? r : 0.0;
        }
        // neighbour-first probing (pk_device.h: curvilinear_search) is exact only where cells cannot overlap
        int probe = desc->neighbour_probe;
        if (const char* e = getenv("PK_NEIGHBOUR_PROBE")) probe = atoi(e);
        if (probe == 0) {
            bool coincident = true;
            std::string msg;
            if (mesh_has_coincident_nodes(ctx->compute, d.node_tab, desc->ny, desc->nx, desc->spherical, d.h_bbox, &coincident, &msg) != hipSuccess)
                return ctx->fail("coincident-node scan: " + msg);
            d.walk_ok = coincident ? 0 : 1;
            if (d.walk_ok) {  // ... and only where the point-in-cell test itself is reliable on every cell
                int* d_flag = nullptr;
                int h_flag = 0;
                PK_HIP(ctx, hipMalloc((void**)&d_flag, sizeof(int)));
                PK_HIP(ctx, hipMemsetAsync(d_flag, 0, sizeof(int), ctx->compute));
                const int64_t ncell = (int64_t)(desc->ny - 1) * desc->nx;
                if (ncell > 0) hipLaunchKernelGGL(illconditioned_cell_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, ctx->compute, d, d_flag);
                PK_HIP(ctx, hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, ctx->compute));
                PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
                (void)hipFree(d_flag);
                if (h_flag) d.walk_ok = 0;
            }
        } else {
            d.walk_ok = probe > 0 ? 1 : 0;
        }
        if (!ctx->no_cell_table && (int64_t)desc->ny * desc->nx < INT32_MAX) {
            // optional: a grid that does not get its table (allocation failure on a crowded device) searches from node_tab
            const int64_t ncell = (int64_t)(desc->ny - 1) * desc->nx;
            double* ct = nullptr;
            const double t0 = now_s();
            if (ncell > 0 && hipMalloc((void**)&ct, (size_t)ncell * CT_STRIDE * sizeof(double)) == hipSuccess) {
                g.allocs.push_back(ct);
                const double t1 = now_s();
                hipLaunchKernelGGL(cell_table_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, ctx->compute, d, ct);
                PK_HIP(ctx, hipGetLastError());
                PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
                d.cell_tab = ct;
                if (pk::print_occupancy())
                    fprintf(stderr, "[pk] cell table: %lld cells, %.2f GB, hipMalloc %.3f s, build %.3f s\n", (long long)ncell,
                            (double)ncell * CT_STRIDE * 8 / 1e9, t1 - t0, now_s() - t1);
            } else {
                (void)hipGetLastError();
            }
        }
        if (!ctx->no_hash_dir) {
            int32_t* dir = nullptr;
            int32_t shift = 0;
            std::string msg;
            if (build_hash_directory(ctx->compute, d.h_keys, d.h_nkeys, &dir, &shift, &msg) != hipSuccess) return ctx->fail(msg);
            if (dir) g.allocs.push_back(dir);
            d.h_dir = dir;
            d.h_dir_shift = shift;
        }
    }
    *grid_id = (int32_t)ctx->grids.size() - 1;
    return 0;
}

int32_t pk_grid_hash_info(pk_ctx* ctx, int32_t grid, pk_hash_info* out) {
    if (!ctx || !out) return -2;
    if (grid < 0 || grid >= (int)ctx->grids.size()) return ctx->fail("unknown grid");
    const HostGrid& g = ctx->grids[grid];
    out->nkeys = g.d.h_nkeys;
    out->nentries = g.h_nentries;
    out->bitwidth = g.d.h_bitwidth;
    out->neighbour_probe = g.d.walk_ok;
    for (int k = 0; k < 6; k++) out->bbox[k] = g.d.h_bbox[k];
    return 0;
}

int32_t pk_grid_hash_download(pk_ctx* ctx, int32_t grid, uint32_t* keys, int64_t* starts, int64_t* counts, uint32_t* faces) {
    if (!ctx) return -2;
    if (grid < 0 || grid >= (int)ctx->grids.size()) return ctx->fail("unknown grid");
    const HostGrid& g = ctx->grids[grid];
    if (g.d.kind != 1 || g.d.h_nkeys <= 0) return ctx->fail("grid has no spatial-hash table");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
    const size_t nk = (size_t)g.d.h_nkeys, ne = (size_t)g.h_nentries;
    if (keys) PK_HIP(ctx, hipMemcpy(keys, g.d.h_keys, nk * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (starts) PK_HIP(ctx, hipMemcpy(starts, g.d.h_starts, nk * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (counts) PK_HIP(ctx, hipMemcpy(counts, g.d.h_counts, nk * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (faces) PK_HIP(ctx, hipMemcpy(faces, g.d.h_faces, ne * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

// ---- fields --------------------------------------------------------------------------------------------
int32_t pk_field_create(pk_ctx* ctx, const pk_field_desc* desc, int32_t* field_id) {
    if (!ctx || !desc || !field_id) return -2;
    if ((int)ctx->fields.size() >= PK_MAX_FIELDS) return ctx->fail("too many fields (PK_MAX_FIELDS)");
    if (desc->grid < 0 || desc->grid >= (int)ctx->grids.size()) return ctx->fail("field refers to an unknown grid");
    if (desc->dtype != PK_F32 && desc->dtype != PK_F64) return ctx->fail("field dtype must be PK_F32 or PK_F64");
    if (desc->nt < 1 || desc->nz < 1 || desc->ny < 1 || desc->nx < 1) return ctx->fail("field extents must be >= 1");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    ctx->fields.emplace_back();
    HostField& f = ctx->fields.back();
    f.desc = *desc;
    const size_t esz = desc->dtype == PK_F64 ? 8 : 4;
    const size_t level_elems = (size_t)desc->nz * desc->ny * desc->nx;
    f.level_bytes = level_elems * esz;
    int nslots = desc->nslots;
    if (nslots <= 0 || nslots >= desc->nt) nslots = desc->nt;
    if (nslots < desc->nt && nslots < 2) nslots = 2;
    f.slot_level.assign(nslots, -1);
    f.slot_pending.assign(nslots, -1);
    f.slot_gen.assign(nslots, 0);
    int ncomp = 1, comp = 0;
    if (desc->pack_leader >= 0) {  // follower: share the leader's interleaved buffer
        if (desc->pack_leader >= (int)ctx->fields.size() - 1) return ctx->fail("pack_leader must be an existing field");
        HostField& L = ctx->fields[desc->pack_leader];
        if (L.d.ncomp <= 1 || L.pack_used >= L.d.ncomp) return ctx->fail("pack leader has no free component");
        if (L.desc.dtype != desc->dtype || L.desc.nt != desc->nt || L.desc.nz != desc->nz || L.desc.ny != desc->ny ||
            L.desc.nx != desc->nx || L.d.nslots != nslots)
            return ctx->fail("packed fields must share dtype, extents and nslots");
        ncomp = L.d.ncomp;
        comp = L.pack_used++;
        f.dev_data = L.dev_data;
        f.owns_data = false;
    } else {
        ncomp = desc->pack_count > 1 ? desc->pack_count : 1;
        PK_HIP(ctx, hipMalloc(&f.dev_data, f.level_bytes * nslots * ncomp));
        PK_HIP(ctx, hipMemset(f.dev_data, 0, f.level_bytes * nslots * ncomp));  // never expose NaN garbage (weight-0 reads)
    }
    if (nslots < desc->nt) {
        // a streamed field: pin the staging chunks now (at creation, not inside the first run), sized for the largest streamed level --
        // at most 3 x 256 MiB (pinning that costs ~0.2 s), a few MB for a small field: locked memory is scarce on shared nodes
        const size_t mib = (size_t)1 << 20;
        const size_t want_raw = f.level_bytes * (size_t)(desc->pack_count > 1 ? desc->pack_count : 1);
        size_t want = std::min(PK_STAGE_CHUNK_BYTES, (want_raw + mib - 1) / mib * mib);
        if (want > ctx->stage_chunk) {
            PK_HIP(ctx, hipStreamSynchronize(ctx->copy));
            for (int k = 0; k < PK_STAGE_BUFFERS; k++)
                if (ctx->stage[k]) { (void)hipHostFree(ctx->stage[k]); ctx->stage[k] = nullptr; }
            ctx->stage_chunk = want;
        }
        for (int k = 0; k < PK_STAGE_BUFFERS; k++)
            if (!ctx->stage[k]) {
                hipError_t e = hipHostMalloc(&ctx->stage[k], ctx->stage_chunk, hipHostMallocDefault);
                while (e != hipSuccess && k == 0 && ctx->stage_chunk > 16 * mib) {  // pinning failed: smaller chunks (more DMA calls, same result)
                    (void)hipGetLastError();
                    ctx->stage_chunk /= 2;
                    e = hipHostMalloc(&ctx->stage[k], ctx->stage_chunk, hipHostMallocDefault);
                }
                if (e != hipSuccess) return ctx->fail("hipHostMalloc of a staging chunk", e);
                ctx->stage_bytes[k] = ctx->stage_chunk;
                PK_HIP(ctx, hipEventRecord(ctx->stage_ev[k], ctx->copy));
            }
    }
    f.time.assign(desc->nt, 0.0);
    if (desc->time) std::copy(desc->time, desc->time + desc->nt, f.time.begin());
    PK_HIP(ctx, hipMalloc((void**)&f.dev_time, sizeof(double) * desc->nt));
    PK_HIP(ctx, hipMemcpy(f.dev_time, f.time.data(), sizeof(double) * desc->nt, hipMemcpyHostToDevice));
    DField& d = f.d;
    memset(&d, 0, sizeof(d));
    d.grid = desc->grid;
    d.dtype = desc->dtype;
    d.nt = desc->nt; d.nz = desc->nz; d.ny = desc->ny; d.nx = desc->nx;
    d.has_time_interval = desc->has_time_interval && desc->nt > 1;
    d.is_const = desc->is_const;
    d.nslots = nslots;
    d.ncomp = ncomp;
    d.comp = comp;
    d.st_t = desc->has_t ? (int64_t)level_elems : 0;
    d.st_z = desc->has_z ? (int64_t)desc->ny * desc->nx : 0;
    d.st_y = desc->has_y ? (int64_t)desc->nx : 0;
    d.st_x = desc->has_x ? 1 : 0;
    d.data = f.dev_data;
    d.time = f.dev_time;
    d.tlen = f.time.back() - f.time.front();
    d.tfirst = f.time.front();
    d.tlast = f.time.back();
    if (level_elems >= (1ull << 31)) return ctx->fail("a field time level must hold < 2^31 elements");
    *field_id = (int32_t)ctx->fields.size() - 1;
    return 0;
}

int32_t pk_field_upload_level(pk_ctx* ctx, int32_t field_id, int32_t level, const void* host_data, int32_t async) {
    if (!ctx || !host_data) return -2;
    if (field_id < 0 || field_id >= (int)ctx->fields.size()) return ctx->fail("unknown field id");
    HostField& f = ctx->fields[field_id];
    if (level < 0 || level >= f.desc.nt) return ctx->fail("time level out of range");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    const int slot = level % f.d.nslots;
    const int ncomp = f.d.ncomp;
    const size_t esz = f.desc.dtype == PK_F64 ? 8 : 4;
    const int64_t level_elems = (int64_t)(f.level_bytes / esz);
    char* dst = (char*)f.dev_data + ((size_t)slot * f.level_bytes * ncomp) + (size_t)f.d.comp * esz;
    char* h2d_dst = dst;
    if (ncomp > 1) {  // packed group: land in a device staging buffer, then interleave on the copy stream
        if (ctx->pack_tmp_bytes < f.level_bytes) {
            PK_HIP(ctx, hipStreamSynchronize(ctx->copy));
            if (ctx->d_pack_tmp) PK_HIP(ctx, hipFree(ctx->d_pack_tmp));
            PK_HIP(ctx, hipMalloc(&ctx->d_pack_tmp, f.level_bytes));
            ctx->pack_tmp_bytes = f.level_bytes;
        }
        h2d_dst = (char*)ctx->d_pack_tmp;
    }
    auto interleave = [&]() -> int32_t {
        if (ncomp <= 1) return 0;
        const unsigned grid = (unsigned)std::min<int64_t>((level_elems + 255) / 256, 256 * 32);
        if (esz == 8) hipLaunchKernelGGL((interleave_kernel<double>), dim3(grid), dim3(256), 0, ctx->copy, (const double*)ctx->d_pack_tmp, (double*)dst, level_elems, ncomp);
        else hipLaunchKernelGGL((interleave_kernel<float>), dim3(grid), dim3(256), 0, ctx->copy, (const float*)ctx->d_pack_tmp, (float*)dst, level_elems, ncomp);
        PK_HIP(ctx, hipGetLastError());
        return 0;
    };
    if (!async) {
        PK_HIP(ctx, hipMemcpyAsync(h2d_dst, host_data, f.level_bytes, hipMemcpyHostToDevice, ctx->copy));
        if (int32_t rc = interleave()) return rc;
        PK_HIP(ctx, hipStreamSynchronize(ctx->copy));
    } else {
        // pageable NumPy memory cannot be DMA'd asynchronously: bounce it through the ring of pinned chunks
        const size_t chunk = ctx->stage_chunk ? ctx->stage_chunk : PK_STAGE_CHUNK_BYTES;
        for (size_t off = 0; off < f.level_bytes; off += chunk) {
            const size_t len = std::min(chunk, f.level_bytes - off);
            int k;
            if (int32_t rc = stage_acquire(ctx, &k)) return rc;
            const double t0 = now_s();
            parallel_memcpy(ctx->stage[k], (const char*)host_data + off, len);
            ctx->stage_fill_s += now_s() - t0;
            if (int32_t rc = stage_submit(ctx, k, h2d_dst + off, len)) return rc;
        }
        if (int32_t rc = interleave()) return rc;
    }
    f.slot_gen[slot] = ++ctx->upload_counter;
    if (async) {  // usable only after pk_field_sync(); the level that lived in this slot is gone as of now
        f.slot_level[slot] = -1;
        f.slot_pending[slot] = level;
        ctx->copy_pending = true;
    } else {
        f.slot_level[slot] = level;
        f.slot_pending[slot] = -1;
    }
    return 0;
}

int32_t pk_field_upload_group_level(pk_ctx* ctx, int32_t leader_id, int32_t level, const void* const* host_data, int32_t ncomp_given,
                                    int32_t async) {
    if (!ctx || !host_data) return -2;
    if (leader_id < 0 || leader_id >= (int)ctx->fields.size()) return ctx->fail("unknown field id");
    HostField& L = ctx->fields[leader_id];
    const int ncomp = L.d.ncomp;
    if (ncomp <= 1 || !L.owns_data) return ctx->fail("pk_field_upload_group_level: the field does not lead a packed group");
    if (ncomp_given != ncomp || L.pack_used != ncomp) return ctx->fail("pk_field_upload_group_level: one host level per component of the complete group is required");
    if (ncomp > 8) return ctx->fail("pk_field_upload_group_level: at most 8 components");
    if (level < 0 || level >= L.desc.nt) return ctx->fail("time level out of range");
    for (int k = 0; k < ncomp; k++)
        if (!host_data[k]) return -2;
    PK_HIP(ctx, hipSetDevice(ctx->device));
    const int slot = level % L.d.nslots;
    const size_t esz = L.desc.dtype == PK_F64 ? 8 : 4;
    const size_t level_elems = L.level_bytes / esz;
    char* dst = (char*)L.dev_data + (size_t)slot * L.level_bytes * ncomp;
    const size_t chunk_elems = (ctx->stage_chunk ? ctx->stage_chunk : PK_STAGE_CHUNK_BYTES) / (esz * ncomp);
    for (size_t off = 0; off < level_elems; off += chunk_elems) {
        const size_t len = std::min(chunk_elems, level_elems - off);
        int k;
        if (int32_t rc = stage_acquire(ctx, &k)) return rc;
        const double t0 = now_s();
        if (esz == 8) {
            const double* src[8];
            for (int c = 0; c < ncomp; c++) src[c] = (const double*)host_data[c] + off;
            parallel_interleave((double*)ctx->stage[k], src, ncomp, len);
        } else {
            const float* src[8];
            for (int c = 0; c < ncomp; c++) src[c] = (const float*)host_data[c] + off;
            parallel_interleave((float*)ctx->stage[k], src, ncomp, len);
        }
        ctx->stage_fill_s += now_s() - t0;
        if (int32_t rc = stage_submit(ctx, k, dst + off * esz * ncomp, len * esz * ncomp)) return rc;
    }
    if (!async) PK_HIP(ctx, hipStreamSynchronize(ctx->copy));
    for (size_t f = 0; f < ctx->fields.size(); f++) {  // the leader and its followers change slot state together
        HostField& F = ctx->fields[f];
        if ((int)f != leader_id && F.desc.pack_leader != leader_id) continue;
        F.slot_gen[slot] = ++ctx->upload_counter;
        F.slot_level[slot] = async ? -1 : level;
        F.slot_pending[slot] = async ? level : -1;
    }
    if (async) ctx->copy_pending = true;
    return 0;
}

int32_t pk_field_sync(pk_ctx* ctx) {
    if (!ctx) return -2;
    PK_HIP(ctx, hipSetDevice(ctx->device));
    PK_HIP(ctx, hipStreamSynchronize(ctx->copy));
    for (auto& f : ctx->fields)
        for (size_t k = 0; k < f.slot_pending.size(); k++)
            if (f.slot_pending[k] >= 0) {
                f.slot_level[k] = f.slot_pending[k];
                f.slot_pending[k] = -1;
            }
    ctx->copy_pending = false;
    return 0;
}

int32_t pk_field_evict_outside(pk_ctx* ctx, int32_t field_id, int32_t lo_level, int32_t hi_level) {
    if (!ctx) return -2;
    if (field_id < 0 || field_id >= (int)ctx->fields.size()) return ctx->fail("unknown field id");
    if (ctx->in_flight) return ctx->fail("pk_field_evict_outside: a launch is in flight");
    HostField& f = ctx->fields[field_id];
    if (f.d.nslots >= f.desc.nt) return 0;  // all levels resident: nothing to evict
    for (size_t k = 0; k < f.slot_level.size(); k++)
        if (f.slot_level[k] >= 0 && (f.slot_level[k] < lo_level || f.slot_level[k] > hi_level)) f.slot_level[k] = -1;
    return 0;
}

int32_t pk_field_slots(pk_ctx* ctx, int32_t field_id, int32_t* levels, int32_t* nslots) {
    if (!ctx) return -2;
    if (field_id < 0 || field_id >= (int)ctx->fields.size()) return ctx->fail("unknown field id");
    HostField& f = ctx->fields[field_id];
    if (nslots) *nslots = f.d.nslots;
    if (levels)
        for (int k = 0; k < f.d.nslots; k++) levels[k] = f.slot_level[k];
    return 0;
}

// ---- particles -----------------------------------------------------------------------------------------
static size_t spatial_size(const pk_ctx* ctx) { return ctx->host.spatial_dtype == PK_F32 ? 4 : 8; }

int32_t pk_particles_bind(pk_ctx* ctx, const pk_particles_desc* host) {
    if (!ctx || !host) return -2;
    if (host->n < 0 || host->ngrids < 1 || host->ngrids > PK_MAX_GRIDS) return ctx->fail("bad particle descriptor");
    if (host->n > 0 && (!host->t || !host->z || !host->y || !host->x || !host->dz || !host->dy || !host->dx || !host->dt ||
                        !host->state || !host->ei || !host->particle_id))
        return ctx->fail("particle columns must not be NULL");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    if (host->n_extra < 0 || host->n_extra > PK_MAX_EXTRA) return ctx->fail("bad number of extra particle columns");
    bool extra_changed = host->n_extra != ctx->host.n_extra;
    for (int k = 0; k < host->n_extra && !extra_changed; k++) extra_changed = host->extra_dtype[k] != ctx->host.extra_dtype[k];
    for (int k = 0; k < host->n_extra; k++) {
        if (host->extra_dtype[k] != PK_F32 && host->extra_dtype[k] != PK_F64) return ctx->fail("extra particle columns must be PK_F32 or PK_F64");
        if (host->n > 0 && !host->extra[k]) return ctx->fail("particle columns must not be NULL");
    }
    const bool realloc_needed = !ctx->bound || host->n > ctx->capacity || host->ngrids != ctx->host.ngrids || extra_changed ||
                                host->spatial_dtype != ctx->host.spatial_dtype || (host->next_dt != nullptr) != (ctx->dev.next_dt != nullptr);
    ctx->host = *host;
    if (realloc_needed) {
        free_particles(ctx);
        const int64_t cap = std::max<int64_t>(host->n, 1);
        const size_t ss = spatial_size(ctx);
        PK_HIP(ctx, hipMalloc((void**)&ctx->dev.t, cap * 8));
        PK_HIP(ctx, hipMalloc(&ctx->dev.z, cap * ss));
        PK_HIP(ctx, hipMalloc(&ctx->dev.y, cap * ss));
        PK_HIP(ctx, hipMalloc(&ctx->dev.x, cap * ss));
        PK_HIP(ctx, hipMalloc(&ctx->dev.dz, cap * ss));
        PK_HIP(ctx, hipMalloc(&ctx->dev.dy, cap * ss));
        PK_HIP(ctx, hipMalloc(&ctx->dev.dx, cap * ss));
        PK_HIP(ctx, hipMalloc((void**)&ctx->dev.dt, cap * 8));
        if (host->next_dt) PK_HIP(ctx, hipMalloc((void**)&ctx->dev.next_dt, cap * 8));
        PK_HIP(ctx, hipMalloc((void**)&ctx->dev.state, cap * 4));
        PK_HIP(ctx, hipMalloc((void**)&ctx->dev.ei, cap * 4 * host->ngrids));
        PK_HIP(ctx, hipMalloc((void**)&ctx->dev.particle_id, cap * 8));
        PK_HIP(ctx, hipMalloc((void**)&ctx->dev.iter, cap * 4));
        for (int k = 0; k < host->n_extra; k++) PK_HIP(ctx, hipMalloc(&ctx->dev.extra[k], cap * (host->extra_dtype[k] == PK_F32 ? 4 : 8)));
        ctx->capacity = cap;
    }
    for (int k = 0; k < PK_MAX_EXTRA; k++) ctx->dev.extra_f32[k] = ctx->alt.extra_f32[k] = (k < host->n_extra && host->extra_dtype[k] == PK_F32);
    ctx->has_perm = false;
    ctx->rerun_valid = false;
    ctx->chk_valid = false;
    ctx->dev.n = host->n;
    ctx->dev.ngrids = host->ngrids;
    ctx->dev.spatial_f32 = host->spatial_dtype == PK_F32;
    ctx->bound = true;
    return 0;
}

}  // extern "C" (helpers with C++ linkage follow)

struct ColRef {
    void* h;
    void* d;
    void* a;  // alt/staging device column
    size_t elem;
    int width;
};

static std::vector<ColRef> particle_columns(pk_ctx* ctx) {
    const size_t ss = spatial_size(ctx);
    const int ng = ctx->host.ngrids;
    return {
        {ctx->host.t, ctx->dev.t, ctx->alt.t, 8, 1},          {ctx->host.z, ctx->dev.z, ctx->alt.z, ss, 1},
        {ctx->host.y, ctx->dev.y, ctx->alt.y, ss, 1},          {ctx->host.x, ctx->dev.x, ctx->alt.x, ss, 1},
        {ctx->host.dz, ctx->dev.dz, ctx->alt.dz, ss, 1},       {ctx->host.dy, ctx->dev.dy, ctx->alt.dy, ss, 1},
        {ctx->host.dx, ctx->dev.dx, ctx->alt.dx, ss, 1},       {ctx->host.dt, ctx->dev.dt, ctx->alt.dt, 8, 1},
        {ctx->host.next_dt, ctx->dev.next_dt, ctx->alt.next_dt, 8, 1},
        {ctx->host.state, ctx->dev.state, ctx->alt.state, 4, 1},
        {ctx->host.ei, ctx->dev.ei, ctx->alt.ei, 4, ng},
        {ctx->host.particle_id, ctx->dev.particle_id, ctx->alt.particle_id, 8, 1},
        {ctx->host.extra[0], ctx->dev.extra[0], ctx->alt.extra[0], (size_t)(ctx->dev.extra_f32[0] ? 4 : 8), 1},
        {ctx->host.extra[1], ctx->dev.extra[1], ctx->alt.extra[1], (size_t)(ctx->dev.extra_f32[1] ? 4 : 8), 1},
        {ctx->host.extra[2], ctx->dev.extra[2], ctx->alt.extra[2], (size_t)(ctx->dev.extra_f32[2] ? 4 : 8), 1},
        {ctx->host.extra[3], ctx->dev.extra[3], ctx->alt.extra[3], (size_t)(ctx->dev.extra_f32[3] ? 4 : 8), 1},
        {ctx->host.extra[4], ctx->dev.extra[4], ctx->alt.extra[4], (size_t)(ctx->dev.extra_f32[4] ? 4 : 8), 1},
        {ctx->host.extra[5], ctx->dev.extra[5], ctx->alt.extra[5], (size_t)(ctx->dev.extra_f32[5] ? 4 : 8), 1},
        {ctx->host.extra[6], ctx->dev.extra[6], ctx->alt.extra[6], (size_t)(ctx->dev.extra_f32[6] ? 4 : 8), 1},
        {ctx->host.extra[7], ctx->dev.extra[7], ctx->alt.extra[7], (size_t)(ctx->dev.extra_f32[7] ? 4 : 8), 1},
        {nullptr, ctx->dev.iter, ctx->alt.iter, 4, 1},  // device only (index PK_NCOLS: beyond the column masks of the ABI)
    };
}
constexpr int PK_NCOLS = 12 + PK_MAX_EXTRA;
static_assert(PK_MAX_EXTRA == 8, "particle_columns lists eight extra columns");

// exchange the two column sets (sizes / flags stay): after the cell sort and the compaction wrote the new order into `alt`
static void swap_column_sets(pk_ctx* ctx) {
    DParticles &d = ctx->dev, &a = ctx->alt;
    std::swap(d.t, a.t); std::swap(d.z, a.z); std::swap(d.y, a.y); std::swap(d.x, a.x);
    std::swap(d.dz, a.dz); std::swap(d.dy, a.dy); std::swap(d.dx, a.dx); std::swap(d.dt, a.dt);
    std::swap(d.next_dt, a.next_dt); std::swap(d.state, a.state); std::swap(d.ei, a.ei); std::swap(d.particle_id, a.particle_id);
    for (int k = 0; k < PK_MAX_EXTRA; k++) std::swap(d.extra[k], a.extra[k]);
    std::swap(d.iter, a.iter);
    ctx->rerun_valid = false;  // the second set no longer holds the state before the last launch
}
// exchange the columns an advection launch writes (pk_device.h: DPOut): the launch read `dev` and wrote `alt`
static void swap_launch_outputs(pk_ctx* ctx) {
    DParticles &d = ctx->dev, &a = ctx->alt;
    std::swap(d.t, a.t); std::swap(d.z, a.z); std::swap(d.y, a.y); std::swap(d.x, a.x);
    std::swap(d.dz, a.dz); std::swap(d.dy, a.dy); std::swap(d.dx, a.dx); std::swap(d.dt, a.dt);
    std::swap(d.next_dt, a.next_dt); std::swap(d.state, a.state); std::swap(d.ei, a.ei); std::swap(d.iter, a.iter);
}

// second column set + permutation buffers, allocated on first use (only when cell sorting is requested)
static int32_t ensure_alt(pk_ctx* ctx) {
    if (ctx->alt.t) return 0;
    const int64_t cap = ctx->capacity;
    const size_t ss = spatial_size(ctx);
    PK_HIP(ctx, hipMalloc((void**)&ctx->alt.t, cap * 8));
    PK_HIP(ctx, hipMalloc(&ctx->alt.z, cap * ss));
    PK_HIP(ctx, hipMalloc(&ctx->alt.y, cap * ss));
    PK_HIP(ctx, hipMalloc(&ctx->alt.x, cap * ss));
    PK_HIP(ctx, hipMalloc(&ctx->alt.dz, cap * ss));
    PK_HIP(ctx, hipMalloc(&ctx->alt.dy, cap * ss));
    PK_HIP(ctx, hipMalloc(&ctx->alt.dx, cap * ss));
    PK_HIP(ctx, hipMalloc((void**)&ctx->alt.dt, cap * 8));
    if (ctx->dev.next_dt) PK_HIP(ctx, hipMalloc((void**)&ctx->alt.next_dt, cap * 8));
    PK_HIP(ctx, hipMalloc((void**)&ctx->alt.state, cap * 4));
    PK_HIP(ctx, hipMalloc((void**)&ctx->alt.ei, cap * 4 * ctx->host.ngrids));
    PK_HIP(ctx, hipMalloc((void**)&ctx->alt.particle_id, cap * 8));
    PK_HIP(ctx, hipMalloc((void**)&ctx->alt.iter, cap * 4));
    for (int k = 0; k < PK_MAX_EXTRA; k++)
        if (ctx->dev.extra[k]) PK_HIP(ctx, hipMalloc(&ctx->alt.extra[k], cap * (ctx->dev.extra_f32[k] ? 4 : 8)));
    PK_HIP(ctx, hipMalloc((void**)&ctx->d_perm, cap * 8));
    PK_HIP(ctx, hipMalloc((void**)&ctx->d_perm_alt, cap * 8));
    PK_HIP(ctx, hipMalloc((void**)&ctx->d_keys, cap * 8));
    PK_HIP(ctx, hipMalloc((void**)&ctx->d_keys_alt, cap * 8));
    PK_HIP(ctx, hipMalloc((void**)&ctx->d_idx, cap * 4));
    PK_HIP(ctx, hipMalloc((void**)&ctx->d_idx_alt, cap * 4));
    return 0;
}

template <class T>
static void launch_gather(pk_ctx* ctx, const void* in, void* out, const uint32_t* perm, int64_t n, int width) {
    hipLaunchKernelGGL((gather_rows_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->compute, (const T*)in, (T*)out, perm, n, width);
}
template <class T>
static void launch_scatter(pk_ctx* ctx, const void* in, void* out, const int64_t* perm, int64_t n, int width) {
    hipLaunchKernelGGL((scatter_rows_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->compute, (const T*)in, (T*)out, perm, n, width);
}

static int bits_for(unsigned long long v) {
    int b = 1;
    while (b < 64 && (v >> b)) b++;
    return b;
}

// Reorder the device rows by cell key (host row order is restored by pk_particles_d2h through d_perm).
static int32_t sort_particles(pk_ctx* ctx, int main_grid, int horizontal_major) {
    const int64_t n = ctx->dev.n;
    if (n < 2) return 0;
    if (n >= (1ll << 32)) return ctx->fail("cell sort supports < 2^32 particles per device");
    int32_t rc = ensure_alt(ctx);
    if (rc) return rc;
    const DGrid& g = ctx->grids[main_grid].d;
    const dim3 grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(sort_key_kernel, grid, dim3(256), 0, ctx->compute, g, ctx->dev, ctx->d_keys, ctx->d_idx, horizontal_major);
    PK_HIP(ctx, hipGetLastError());
    unsigned long long kmax;
    const unsigned long long nzc = (g.has_z && g.nz >= 2) ? (unsigned long long)g.nz : 1ull;
    if (g.kind == 1) kmax = horizontal_major ? ((0x3FFFFFFFull << 12) | 0xFFFull) : ((nzc << 30) | 0x3FFFFFFFull);
    else kmax = nzc * (unsigned long long)std::max(g.ny, 1) * (unsigned long long)std::max(g.nx, 1);
    const unsigned end_bit = (unsigned)bits_for(kmax);
    size_t tmp_bytes = 0;
    PK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, ctx->d_keys, ctx->d_keys_alt, ctx->d_idx, ctx->d_idx_alt, (size_t)n, 0u, end_bit, ctx->compute));
    if (tmp_bytes > ctx->sort_tmp_bytes) {
        if (ctx->d_sort_tmp) PK_HIP(ctx, hipFree(ctx->d_sort_tmp));
        PK_HIP(ctx, hipMalloc(&ctx->d_sort_tmp, tmp_bytes));
        ctx->sort_tmp_bytes = tmp_bytes;
    }
    PK_HIP(ctx, rocprim::radix_sort_pairs(ctx->d_sort_tmp, tmp_bytes, ctx->d_keys, ctx->d_keys_alt, ctx->d_idx, ctx->d_idx_alt, (size_t)n, 0u, end_bit, ctx->compute));
    const uint32_t* perm = ctx->d_idx_alt;
    for (const ColRef& c : particle_columns(ctx)) {
        if (!c.d || !c.a) continue;
        if (c.elem == 8) launch_gather<unsigned long long>(ctx, c.d, c.a, perm, n, c.width);
        else launch_gather<uint32_t>(ctx, c.d, c.a, perm, n, c.width);
    }
    hipLaunchKernelGGL(compose_perm_kernel, grid, dim3(256), 0, ctx->compute, ctx->has_perm ? ctx->d_perm : nullptr, perm, ctx->d_perm_alt, n);
    PK_HIP(ctx, hipGetLastError());
    swap_column_sets(ctx);
    std::swap(ctx->d_perm, ctx->d_perm_alt);
    ctx->has_perm = true;
    return 0;
}

static int32_t copy_particles(pk_ctx* ctx, bool to_device, uint32_t mask = 0xFFFFFFFFu) {
    if (!ctx->bound) return ctx->fail("no particles bound");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n = ctx->host.n;
    ctx->rerun_valid = false;  // d2h stages through the second column set, h2d replaces the state
    if (n == 0) return 0;
    if (to_device) ctx->has_perm = false;  // host order
    int bit = 0;
    for (const ColRef& c : particle_columns(ctx)) {
        const bool selected = (mask >> bit++) & 1u;
        if (!c.h || !c.d || !selected) continue;
        const size_t bytes = (size_t)n * c.elem * c.width;
        if (to_device) {
            PK_HIP(ctx, hipMemcpyAsync(c.d, c.h, bytes, hipMemcpyHostToDevice, ctx->compute));
        } else if (ctx->has_perm) {  // undo the cell sort: host row perm[i] <- device row i
            if (c.elem == 8) launch_scatter<unsigned long long>(ctx, c.d, c.a, ctx->d_perm, n, c.width);
            else launch_scatter<uint32_t>(ctx, c.d, c.a, ctx->d_perm, n, c.width);
            PK_HIP(ctx, hipMemcpyAsync(c.h, c.a, bytes, hipMemcpyDeviceToHost, ctx->compute));
        } else {
            PK_HIP(ctx, hipMemcpyAsync(c.h, c.d, bytes, hipMemcpyDeviceToHost, ctx->compute));
        }
    }
    PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
    return 0;
}

#include "pk_select.inc"

extern "C" {

int32_t pk_particles_h2d(pk_ctx* ctx) {
    if (!ctx) return -2;
    return copy_particles(ctx, true);
}
int32_t pk_particles_d2h(pk_ctx* ctx) {
    if (!ctx) return -2;
    return copy_particles(ctx, false);
}
int32_t pk_particles_set_mask(pk_ctx* ctx, const int32_t* mask) {
    if (!ctx || !mask) return -2;
    if (!ctx->bound) return ctx->fail("no particles bound");
    if (ctx->has_perm) return ctx->fail("pk_particles_set_mask: the device rows are cell-sorted (body_only launches run on host-ordered rows: pk_particles_h2d first)");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->dev.n > 0) PK_HIP(ctx, hipMemcpyAsync(ctx->dev.iter, mask, (size_t)ctx->dev.n * 4, hipMemcpyHostToDevice, ctx->compute));
    PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
    return 0;
}

// The state of every device column (and the row order of the cell sort) as of now, kept on the device: what a Kernel.execute that
// spans several launches (streamed levels, re-sort horizons) goes back to when a particle errs in a later launch and the whole
// batch has to stop at that iteration (kernel.py:236-245).  One packed buffer of ~88 B per particle, allocated on first use.
static int32_t checkpoint_copy(pk_ctx* ctx, bool save) {
    const int64_t n = save ? ctx->dev.n : ctx->chk_n;
    size_t off = 0;
    for (const ColRef& c : particle_columns(ctx)) {
        if (!c.d) continue;
        const size_t bytes = (size_t)n * c.elem * c.width;
        if (bytes) {
            if (save) PK_HIP(ctx, hipMemcpyAsync(ctx->d_chk + off, c.d, bytes, hipMemcpyDeviceToDevice, ctx->compute));
            else PK_HIP(ctx, hipMemcpyAsync(c.d, ctx->d_chk + off, bytes, hipMemcpyDeviceToDevice, ctx->compute));
        }
        off += (bytes + 255) & ~(size_t)255;
    }
    if (save ? ctx->has_perm : ctx->chk_has_perm) {
        const size_t bytes = (size_t)n * 8;
        if (save) PK_HIP(ctx, hipMemcpyAsync(ctx->d_chk + off, ctx->d_perm, bytes, hipMemcpyDeviceToDevice, ctx->compute));
        else PK_HIP(ctx, hipMemcpyAsync(ctx->d_perm, ctx->d_chk + off, bytes, hipMemcpyDeviceToDevice, ctx->compute));
    }
    PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
    return 0;
}
int32_t pk_particles_checkpoint(pk_ctx* ctx) {
    if (!ctx) return -2;
    if (!ctx->bound) return ctx->fail("no particles bound");
    if (ctx->in_flight) return ctx->fail("pk_particles_checkpoint: a launch is in flight (call pk_execute_end)");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n = ctx->dev.n;
    size_t need = 0;
    for (const ColRef& c : particle_columns(ctx))
        if (c.d) need += ((size_t)n * c.elem * c.width + 255) & ~(size_t)255;
    need += (size_t)n * 8 + 256;  // the permutation
    if (need > ctx->chk_bytes) {
        if (ctx->d_chk) (void)hipFree(ctx->d_chk);
        ctx->d_chk = nullptr;
        ctx->chk_bytes = 0;
        PK_HIP(ctx, hipMalloc((void**)&ctx->d_chk, need));
        ctx->chk_bytes = need;
    }
    ctx->chk_valid = false;
    const int32_t rc = checkpoint_copy(ctx, true);
    if (rc) return rc;
    ctx->chk_n = n;
    ctx->chk_has_perm = ctx->has_perm;
    ctx->chk_valid = true;
    return 0;
}
int32_t pk_particles_restore(pk_ctx* ctx) {
    if (!ctx) return -2;
    if (!ctx->bound) return ctx->fail("no particles bound");
    if (ctx->in_flight) return ctx->fail("pk_particles_restore: a launch is in flight (call pk_execute_end)");
    if (!ctx->chk_valid) return ctx->fail("pk_particles_restore: no checkpoint (pk_particles_checkpoint; binding new particles discards it)");
    if (ctx->chk_n != ctx->dev.n) return ctx->fail("pk_particles_restore: the particle count changed since the checkpoint");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->chk_has_perm && !ctx->d_perm) return ctx->fail("pk_particles_restore: permutation buffer missing (internal error)");
    const int32_t rc = checkpoint_copy(ctx, false);
    if (rc) return rc;
    ctx->has_perm = ctx->chk_has_perm;
    ctx->rerun_valid = false;
    return 0;
}

int32_t pk_particles_d2h_columns(pk_ctx* ctx, uint32_t mask) {
    if (!ctx) return -2;
    return copy_particles(ctx, false, mask);
}

int32_t pk_particles_h2d_columns(pk_ctx* ctx, uint32_t mask) {
    if (!ctx) return -2;
    if (!ctx->bound) return ctx->fail("no particles bound");
    if (ctx->in_flight) return ctx->fail("pk_particles_h2d_columns: a launch is in flight (call pk_execute_end)");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n = ctx->host.n;
    ctx->rerun_valid = false;  // the staging below goes through the second column set
    ctx->chk_valid = false;    // a checkpoint no longer describes these columns
    if (n == 0) return 0;
    if (ctx->has_perm) {
        const int32_t rc = ensure_alt(ctx);
        if (rc) return rc;
    }
    int bit = 0;
    for (const ColRef& c : particle_columns(ctx)) {
        const bool selected = (mask >> bit++) & 1u;
        if (!c.h || !c.d || !selected) continue;
        const size_t bytes = (size_t)n * c.elem * c.width;
        if (!ctx->has_perm) {
            PK_HIP(ctx, hipMemcpyAsync(c.d, c.h, bytes, hipMemcpyHostToDevice, ctx->compute));
        } else {  // device row i holds host row perm[i]
            PK_HIP(ctx, hipMemcpyAsync(c.a, c.h, bytes, hipMemcpyHostToDevice, ctx->compute));
            const dim3 grid((unsigned)((n + 255) / 256));
            if (c.elem == 8) hipLaunchKernelGGL((gather_rows64_kernel<unsigned long long>), grid, dim3(256), 0, ctx->compute, (const unsigned long long*)c.a, (unsigned long long*)c.d, ctx->d_perm, n, c.width);
            else hipLaunchKernelGGL((gather_rows64_kernel<uint32_t>), grid, dim3(256), 0, ctx->compute, (const uint32_t*)c.a, (uint32_t*)c.d, ctx->d_perm, n, c.width);
            PK_HIP(ctx, hipGetLastError());
        }
    }
    PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
    return 0;
}

int32_t pk_particles_fill_f64(pk_ctx* ctx, uint32_t mask, double value) {
    if (!ctx) return -2;
    if (!ctx->bound) return ctx->fail("no particles bound");
    if (ctx->in_flight) return ctx->fail("pk_particles_fill_f64: a launch is in flight (call pk_execute_end)");
    if (mask & ~(PK_COL_T | PK_COL_DT | PK_COL_NEXT_DT)) return ctx->fail("pk_particles_fill_f64: float64 columns only (t, dt, next_dt)");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n = ctx->dev.n;
    ctx->rerun_valid = false;
    ctx->chk_valid = false;
    if (n == 0) return 0;
    const dim3 grid((unsigned)((n + 255) / 256));
    double* cols[3] = {(mask & PK_COL_T) ? ctx->dev.t : nullptr, (mask & PK_COL_DT) ? ctx->dev.dt : nullptr, (mask & PK_COL_NEXT_DT) ? ctx->dev.next_dt : nullptr};
    for (double* c : cols)
        if (c) hipLaunchKernelGGL(fill_f64_kernel, grid, dim3(256), 0, ctx->compute, c, n, value);
    PK_HIP(ctx, hipGetLastError());
    return 0;  // (stream order: the next launch sees it)
}

int32_t pk_particles_t_stats(pk_ctx* ctx, double* t_min, double* t_max, int64_t* n_nan) {
    if (!ctx || !t_min || !t_max || !n_nan) return -2;
    if (!ctx->bound) return ctx->fail("no particles bound");
    if (ctx->in_flight) return ctx->fail("pk_particles_t_stats: a launch is in flight (call pk_execute_end)");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n = ctx->dev.n;
    *t_min = *t_max = NAN;
    *n_nan = 0;
    if (n == 0) return 0;
    const unsigned long long init[3] = {~0ull, 0ull, 0ull};
    PK_HIP(ctx, hipMemcpyAsync(ctx->d_tstats, init, sizeof(init), hipMemcpyHostToDevice, ctx->compute));
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(t_stats_kernel, dim3(grid), dim3(256), 0, ctx->compute, ctx->dev.t, n, ctx->d_tstats);
    PK_HIP(ctx, hipGetLastError());
    unsigned long long got[3];
    PK_HIP(ctx, hipMemcpyAsync(got, ctx->d_tstats, sizeof(got), hipMemcpyDeviceToHost, ctx->compute));
    PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
    auto unmap = [](unsigned long long b) {
        b = (b & 0x8000000000000000ull) ? (b & 0x7fffffffffffffffull) : ~b;
        double v;
        memcpy(&v, &b, 8);
        return v;
    };
    if (got[0] != ~0ull || got[1] != 0ull) {
        *t_min = unmap(got[0]);
        *t_max = unmap(got[1]);
    }
    *n_nan = (int64_t)got[2];
    return 0;
}

// ---- asynchronous write-out (particleset.py:436-459 / particlefile.py:142-180 overlapped with the next interval) -------------
static int32_t snapshot_begin(pk_ctx* ctx, uint32_t mask, int32_t slot, bool filtered, double t_out);
int32_t pk_particles_snapshot_begin(pk_ctx* ctx, uint32_t mask, int32_t slot) { return snapshot_begin(ctx, mask, slot, false, 0.0); }
int32_t pk_particles_snapshot_filtered(pk_ctx* ctx, uint32_t mask, int32_t slot, double t) { return snapshot_begin(ctx, mask, slot, true, t); }
}  // extern "C"
// filtered: only the rows that pass the write filter of particlefile.py:198-221 at output time t_out, packed in host row order (pk_select.inc)
static int32_t snapshot_begin(pk_ctx* ctx, uint32_t mask, int32_t slot, bool filtered, double t_out) {
    if (!ctx) return -2;
    if (slot < 0 || slot > 1) return ctx->fail("snapshot slot must be 0 or 1");
    if (!ctx->bound) return ctx->fail("no particles bound");
    if (ctx->in_flight) return ctx->fail("pk_particles_snapshot_begin: a launch is in flight");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    pk_ctx::Snapshot& sn = ctx->snap[slot];
    if (sn.in_flight) return ctx->fail("snapshot slot still in flight (pk_particles_snapshot_wait first)");
    if (!sn.ready) {
        PK_HIP(ctx, hipEventCreateWithFlags(&sn.ready, hipEventDisableTiming));
        PK_HIP(ctx, hipEventCreateWithFlags(&sn.done, hipEventDisableTiming));
    }
    const int64_t n = ctx->dev.n;
    const std::vector<ColRef> cols = particle_columns(ctx);
    const size_t ss = spatial_size(ctx);
    bool fits = sn.capacity >= n && sn.ngrids == ctx->host.ngrids && sn.ss == ss;
    for (int k = 0; k < PK_MAX_EXTRA; k++) fits = fits && sn.extra_elem[k] == (cols[12 + k].d ? cols[12 + k].elem : 0);
    for (int k = 0; k < PK_NCOLS; k++)
        if (((mask >> k) & 1u) && cols[k].d && !sn.dev[k]) fits = false;  // a column the buffers were not sized for
    if (!fits) {  // (re)size the columns of this mask only: pinning costs ~0.3 s per GB
        for (int k = 0; k < PK_NCOLS; k++) {
            if (sn.dev[k]) PK_HIP(ctx, hipFree(sn.dev[k]));
            if (sn.host[k]) PK_HIP(ctx, hipHostFree(sn.host[k]));
            sn.dev[k] = sn.host[k] = nullptr;
        }
        const int64_t cap = std::max<int64_t>(n, 1);  // particle sets only shrink during a run (deletions)
        for (int k = 0; k < PK_NCOLS; k++) {
            if (!((mask >> k) & 1u) || !cols[k].d) continue;
            const size_t bytes = (size_t)cap * cols[k].elem * cols[k].width;
            PK_HIP(ctx, hipMalloc(&sn.dev[k], bytes));
            PK_HIP(ctx, hipHostMalloc(&sn.host[k], bytes, hipHostMallocDefault));
        }
        sn.capacity = cap;
        sn.ngrids = ctx->host.ngrids;
        sn.ss = ss;
        for (int k = 0; k < PK_MAX_EXTRA; k++) sn.extra_elem[k] = cols[12 + k].d ? cols[12 + k].elem : 0;
    }
    sn.n = n;
    sn.mask = mask;
    int64_t n_out = n;
    if (filtered && n > 0) {
        int32_t rc = select_flags(ctx, t_out, 1, &n_out);  // (one 8-byte read-back: the row count the D2H below needs)
        if (rc) return rc;
        for (int k = 0; k < PK_NCOLS && n_out > 0; k++) {
            const ColRef& c = cols[k];
            if (!((mask >> k) & 1u) || !c.d) continue;
            select_column(ctx, c, sn.dev[k]);
        }
        PK_HIP(ctx, hipGetLastError());
        sn.n = n_out;
    } else if (n > 0) {
        for (int k = 0; k < PK_NCOLS; k++) {
            const ColRef& c = cols[k];
            if (!((mask >> k) & 1u) || !c.d) continue;
            const size_t bytes = (size_t)n * c.elem * c.width;
            if (ctx->has_perm) {  // undo the cell sort: host row perm[i] <- device row i
                if (c.elem == 8) launch_scatter<unsigned long long>(ctx, c.d, sn.dev[k], ctx->d_perm, n, c.width);
                else launch_scatter<uint32_t>(ctx, c.d, sn.dev[k], ctx->d_perm, n, c.width);
            } else {
                PK_HIP(ctx, hipMemcpyAsync(sn.dev[k], c.d, bytes, hipMemcpyDeviceToDevice, ctx->compute));
            }
        }
        PK_HIP(ctx, hipGetLastError());
    }
    PK_HIP(ctx, hipEventRecord(sn.ready, ctx->compute));  // the next launch may now overwrite the live columns
    PK_HIP(ctx, hipStreamWaitEvent(ctx->copy, sn.ready, 0));
    for (int k = 0; k < PK_NCOLS && n_out > 0; k++) {
        const ColRef& c = cols[k];
        if (!((mask >> k) & 1u) || !c.d) continue;
        PK_HIP(ctx, hipMemcpyAsync(sn.host[k], sn.dev[k], (size_t)n_out * c.elem * c.width, hipMemcpyDeviceToHost, ctx->copy));
    }
    PK_HIP(ctx, hipEventRecord(sn.done, ctx->copy));
    sn.in_flight = true;
    return 0;
}
extern "C" {

// Wait for the snapshot in `slot` and hand out its pinned host columns (valid until the next snapshot_begin on that slot).
// May be called from a second host thread while the first one drives the next launch: it only waits on an event.
int32_t pk_particles_snapshot_wait(pk_ctx* ctx, int32_t slot, pk_particles_desc* out) {
    if (!ctx || !out) return -2;
    if (slot < 0 || slot > 1) return -2;
    pk_ctx::Snapshot& sn = ctx->snap[slot];
    if (!sn.in_flight) return -3;  // (no ctx->fail here: the error string belongs to the driving thread)
    if (hipSetDevice(ctx->device) != hipSuccess) return -1;
    if (hipEventSynchronize(sn.done) != hipSuccess) return -1;
    sn.in_flight = false;
    memset(out, 0, sizeof(*out));
    out->n = sn.n;
    out->ngrids = sn.ngrids;
    out->spatial_dtype = sn.ss == 4 ? PK_F32 : PK_F64;
    void** dst[12] = {(void**)&out->t, &out->z, &out->y, &out->x, &out->dz, &out->dy, &out->dx, (void**)&out->dt, (void**)&out->next_dt,
                      (void**)&out->state, (void**)&out->ei, (void**)&out->particle_id};
    for (int k = 0; k < 12; k++) *dst[k] = ((sn.mask >> k) & 1u) ? sn.host[k] : nullptr;
    for (int k = 0; k < PK_MAX_EXTRA; k++) {
        out->extra[k] = ((sn.mask >> (12 + k)) & 1u) ? sn.host[12 + k] : nullptr;
        out->extra_dtype[k] = sn.extra_elem[k] == 4 ? PK_F32 : PK_F64;
        if (sn.extra_elem[k]) out->n_extra = k + 1;
    }
    return 0;
}

int32_t pk_particles_device(pk_ctx* ctx, pk_particles_desc* dev, int64_t** perm) {
    if (!ctx || !dev) return -2;
    if (!ctx->bound) return ctx->fail("no particles bound");
    dev->n = ctx->dev.n;
    dev->ngrids = ctx->dev.ngrids;
    dev->spatial_dtype = ctx->host.spatial_dtype;
    dev->t = ctx->dev.t;
    dev->z = ctx->dev.z; dev->y = ctx->dev.y; dev->x = ctx->dev.x;
    dev->dz = ctx->dev.dz; dev->dy = ctx->dev.dy; dev->dx = ctx->dev.dx;
    dev->dt = ctx->dev.dt;
    dev->next_dt = ctx->dev.next_dt;
    dev->state = ctx->dev.state;
    dev->ei = ctx->dev.ei;
    dev->particle_id = ctx->dev.particle_id;
    if (perm) *perm = ctx->has_perm ? ctx->d_perm : nullptr;
    return 0;
}

// Remove the rows whose state is Delete from the DEVICE-RESIDENT columns (order of the survivors kept, in device order and
// in host order) and re-point the host side at `new_host`, the caller's arrays of the surviving length: no column crosses
// PCIe.  Kernel.remove_deleted (kernel.py:98-106) without the round trip through NumPy.
int32_t pk_particles_compact(pk_ctx* ctx, const pk_particles_desc* new_host, int64_t* n_new) {
    if (!ctx || !new_host || !n_new) return -2;
    if (!ctx->bound) return ctx->fail("no particles bound");
    if (ctx->in_flight) return ctx->fail("pk_particles_compact: a launch is in flight");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n = ctx->dev.n;
    if (n >= (1ll << 32)) return ctx->fail("device compaction supports < 2^32 particles per device");
    if (new_host->ngrids != ctx->host.ngrids || new_host->spatial_dtype != ctx->host.spatial_dtype ||
        (new_host->next_dt != nullptr) != (ctx->host.next_dt != nullptr) || new_host->n_extra != ctx->host.n_extra)
        return ctx->fail("pk_particles_compact: the new host columns must have the bound schema");
    int64_t kept = 0;
    if (n > 0) {
        int32_t rc = ensure_alt(ctx);
        if (rc) return rc;
        const dim3 grid((unsigned)((n + 255) / 256));
        unsigned long long *keep = ctx->d_keys, *pos = ctx->d_keys_alt;
        uint32_t *keep_host = ctx->d_idx, *host_pos = ctx->d_idx_alt;
        const int64_t* perm = ctx->has_perm ? ctx->d_perm : nullptr;
        hipLaunchKernelGGL(keep_flags_kernel, grid, dim3(256), 0, ctx->compute, ctx->dev.state, perm, n, keep, keep_host);
        size_t tb1 = 0, tb2 = 0;
        PK_HIP(ctx, rocprim::exclusive_scan(nullptr, tb1, keep, pos, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), ctx->compute));
        PK_HIP(ctx, rocprim::exclusive_scan(nullptr, tb2, keep_host, host_pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->compute));
        const size_t tb = std::max(tb1, tb2);
        if (tb > ctx->sort_tmp_bytes) {
            if (ctx->d_sort_tmp) PK_HIP(ctx, hipFree(ctx->d_sort_tmp));
            PK_HIP(ctx, hipMalloc(&ctx->d_sort_tmp, tb));
            ctx->sort_tmp_bytes = tb;
        }
        PK_HIP(ctx, rocprim::exclusive_scan(ctx->d_sort_tmp, tb1, keep, pos, 0ull, (size_t)n, rocprim::plus<unsigned long long>(), ctx->compute));
        PK_HIP(ctx, rocprim::exclusive_scan(ctx->d_sort_tmp, tb2, keep_host, host_pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), ctx->compute));
        unsigned long long last_pos = 0, last_keep = 0;
        PK_HIP(ctx, hipMemcpyAsync(&last_pos, pos + (n - 1), 8, hipMemcpyDeviceToHost, ctx->compute));
        PK_HIP(ctx, hipMemcpyAsync(&last_keep, keep + (n - 1), 8, hipMemcpyDeviceToHost, ctx->compute));
        for (const ColRef& c : particle_columns(ctx)) {
            if (!c.d || !c.a) continue;
            if (c.elem == 8) hipLaunchKernelGGL((compact_rows_kernel<unsigned long long>), grid, dim3(256), 0, ctx->compute, (const unsigned long long*)c.d, (unsigned long long*)c.a, keep, pos, n, c.width);
            else hipLaunchKernelGGL((compact_rows_kernel<uint32_t>), grid, dim3(256), 0, ctx->compute, (const uint32_t*)c.d, (uint32_t*)c.a, keep, pos, n, c.width);
        }
        if (perm) hipLaunchKernelGGL(compact_perm_kernel, grid, dim3(256), 0, ctx->compute, perm, keep, pos, host_pos, n, ctx->d_perm_alt);
        PK_HIP(ctx, hipGetLastError());
        PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
        kept = (int64_t)(last_pos + last_keep);
        swap_column_sets(ctx);
        if (perm) std::swap(ctx->d_perm, ctx->d_perm_alt);
    }
    if (new_host->n != kept) return ctx->fail("pk_particles_compact: new host columns hold " + std::to_string(new_host->n) + " rows, " +
                                              std::to_string(kept) + " particles survive");
    ctx->dev.n = kept;
    ctx->host = *new_host;
    *n_new = kept;
    return 0;
}

// ---- execution -----------------------------------------------------------------------------------------
// Some grid stores a coordinate as float32: NumPy's float32 arithmetic on coordinate / barycentric arrays has to be reproduced
// (pk_device.h: TYPED), which only the PROG_TYPED program and the typed sampling kernels carry.
static bool ctx_is_typed(const pk_ctx* ctx) {
    for (const HostGrid& g : ctx->grids)
        if (g.d.lon_f32 || g.d.lat_f32 || g.d.depth_f32) return true;
    return false;
}

static int32_t fill_args(pk_ctx* ctx, const pk_exec_params* prm, KArgs& a, size_t& lds_bytes, int& use_lds) {
    memset(&a, 0, sizeof(a));
    {
        // descriptor tables: [grids][fields] in one device buffer, refreshed on the compute stream when anything changed since the last
        // launch (ring slots move with every streamed level).  In stream order, so a kernel still running keeps reading the old values.
        const size_t gb = ctx->grids.size() * sizeof(DGrid), fb = ctx->fields.size() * sizeof(DField);
        const size_t need = gb + fb;
        if (need > ctx->desc_bytes) {
            const size_t cap = (size_t)PK_MAX_GRIDS * sizeof(DGrid) + (size_t)PK_MAX_FIELDS * sizeof(DField);
            if (ctx->d_desc) (void)hipFree(ctx->d_desc);
            if (ctx->h_desc) (void)hipHostFree(ctx->h_desc);
            ctx->d_desc = ctx->h_desc = nullptr;
            ctx->desc_bytes = 0;
            PK_HIP(ctx, hipMalloc((void**)&ctx->d_desc, cap));
            PK_HIP(ctx, hipHostMalloc((void**)&ctx->h_desc, 2 * cap, hipHostMallocDefault));  // staged copy + what the device holds
            memset(ctx->h_desc, 0xFF, 2 * cap);
            ctx->desc_bytes = cap;
        }
        char* stage = ctx->h_desc;                    // what this launch needs
        char* mirror = ctx->h_desc + ctx->desc_bytes;  // what was uploaded last
        std::vector<char> cur(need);
        for (size_t g = 0; g < ctx->grids.size(); g++) memcpy(cur.data() + g * sizeof(DGrid), &ctx->grids[g].d, sizeof(DGrid));
        for (size_t f = 0; f < ctx->fields.size(); f++) memcpy(cur.data() + gb + f * sizeof(DField), &ctx->fields[f].d, sizeof(DField));
        if (need && memcmp(cur.data(), mirror, need) != 0) {
            // the previous upload from `stage` completed long ago unless a launch is still queued behind it: wait for the stream then
            PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
            memcpy(stage, cur.data(), need);
            PK_HIP(ctx, hipMemcpyAsync(ctx->d_desc, stage, need, hipMemcpyHostToDevice, ctx->compute));
            memcpy(mirror, cur.data(), need);
        }
        a.grids = (const PK_CONST_AS DGrid*)ctx->d_desc;
        a.fields = (const PK_CONST_AS DField*)(ctx->d_desc + gb);
    }
    a.p = ctx->dev;
    a.prm = *prm;
    a.counters = ctx->d_counters;
    a.twe_listed = ctx->d_twe;
    a.twe_found = ctx->d_twe_found;
    a.twe_hit = ctx->d_twe_hit;
    const int nf = (int)ctx->fields.size();
    auto valid = [&](int f) { return f >= 0 && f < nf; };
    if (!valid(prm->fU) || !valid(prm->fV)) return ctx->fail("params.fU/fV must name existing fields");
    if (prm->fW >= nf || prm->fKh_zonal >= nf || prm->fKh_meridional >= nf) return ctx->fail("params refer to unknown fields");
    a.main_field = prm->fU;
    a.main_grid = ctx->fields[prm->fU].d.grid;
    const HostField& mf = ctx->fields[prm->fU];
    const HostGrid& mg = ctx->grids[a.main_grid];
    // resident time window: the intersection over every time-varying field that lives in a ring of slots (each on its own
    // time axis).  A particle steps only while [t, t+dt] lies inside it, so no field is sampled on a level that is not there.
    a.win_lo = -INFINITY;
    a.win_hi = INFINITY;
    for (const HostField& f : ctx->fields) {
        if (!f.d.has_time_interval || f.d.nslots >= f.desc.nt) continue;
        int lo = 1 << 30, hi = -1, cnt = 0;
        for (int lv : f.slot_level)
            if (lv >= 0) { lo = std::min(lo, lv); hi = std::max(hi, lv); cnt++; }
        if (cnt == 0) return ctx->fail("no time level of a streamed field is resident (pk_field_upload_level)");
        if (hi - lo + 1 != cnt) return ctx->fail("resident time levels are not contiguous (pk_field_evict_outside)");
        // nothing earlier / later exists at the ends of the axis: let the kernels raise code 70 there
        if (lo > 0) a.win_lo = std::max(a.win_lo, f.time[lo]);
        if (hi < f.desc.nt - 1) a.win_hi = std::min(a.win_hi, f.time[hi]);
    }
    // soft horizon of the launch (re-sort cadence of a long fused run): same pause as at the edge of a level ring
    if (prm->horizon_lo < prm->horizon_hi) {  // (lo >= hi, e.g. a zeroed struct: no horizon)
        if (prm->horizon_lo > a.win_lo) a.win_lo = prm->horizon_lo;
        if (prm->horizon_hi < a.win_hi) a.win_hi = prm->horizon_hi;
    }
    // LDS staging of the main grid's 1-D vectors
    const int nt = mf.d.has_time_interval ? mf.d.nt : 0;
    const int nz = mg.d.has_z ? mg.d.nz : 0;
    const int ny = mg.d.kind == 0 ? mg.d.ny : 0;
    const int nx = mg.d.kind == 0 ? mg.d.nx : 0;
    a.lds_time = 0;
    a.lds_depth = a.lds_time + nt;
    a.lds_lat = a.lds_depth + nz;
    a.lds_lon = a.lds_lat + ny;
    a.lds_total = a.lds_lon + nx;
    lds_bytes = (size_t)std::max(a.lds_total, 1) * sizeof(double);
    use_lds = lds_bytes <= 64 * 1024;
    if (!use_lds) lds_bytes = 0;
    // per-lane cell cache of curvilinear grids (pk_device.h: CellCache): CC_NODE_ROWS doubles + 4 key ints per lane, plus the
    // 12 staggered field values of a C-grid evaluation when they still fit the 64 KiB of a workgroup
    a.lds_cc_nodes = a.lds_cc_keys = a.lds_cc_fvals = -1;
    const bool want_cc = use_lds && mg.d.kind == 1 && !ctx->no_cell_cache && (int64_t)mg.d.ny * mg.d.nx < INT32_MAX;
    if (want_cc) {
        const size_t node_b = CC_NODE_ROWS * CC_LANES * sizeof(double), key_b = 4 * CC_LANES * sizeof(int32_t);
        const size_t fval_b = 12 * CC_LANES * (size_t)(mf.desc.dtype == PK_F32 ? 4 : 8);
        if (lds_bytes + node_b + key_b <= 64 * 1024) {
            a.lds_cc_nodes = a.lds_total;
            a.lds_cc_keys = a.lds_cc_nodes + CC_NODE_ROWS * CC_LANES;
            int32_t end = a.lds_cc_keys + (int32_t)(key_b / sizeof(double));
            // (A/B on config 3: without the cached field values the RK4_3D launch takes 96.6 ms instead of 66.3 ms)
            if (prm->interp_uv == 1 && (size_t)end * sizeof(double) + fval_b <= 64 * 1024) {
                a.lds_cc_fvals = end;
                end += (int32_t)(fval_b / sizeof(double));
            }
            lds_bytes = (size_t)end * sizeof(double);
        }
    }
    return 0;
}

// Fold the descriptors of the velocity fields and their grid into the wave-uniform constants of the fast A-grid path
// (pk_device.h: FastA, pk_fast_agrid.h) when its preconditions hold: rectilinear grid with float64 coordinates, XLinear_Velocity,
// U / V (/ W) plain arrays of one layout with adjacent x-corners, a time level below 4 GiB (32-bit lane offsets), and coordinate
// vectors whose cell widths have well-scaled reciprocals.  a.fast.ok == 0 otherwise (the general program runs).
static int32_t fill_fast(pk_ctx* ctx, const pk_exec_params* prm, KArgs& a, bool want_w) {
    FastA& F = a.fast;
    memset(&F, 0, sizeof(F));
    if (ctx->no_fast || prm->interp_uv != 0 || prm->rk45_mode) return 0;  // rk45_mode: dt follows the next_dt column (kernel.py:118-120)
    const HostField& U = ctx->fields[prm->fU];
    const HostField& V = ctx->fields[prm->fV];
    const HostField* W = (want_w && prm->fW >= 0) ? &ctx->fields[prm->fW] : nullptr;
    if (want_w && !W) return 0;
    const HostGrid& g = ctx->grids[U.d.grid];
    if (g.d.kind != 0 || g.d.lon_f32 || g.d.lat_f32 || g.d.depth_f32) return 0;
    if (V.d.grid != U.d.grid || (W && W->d.grid != U.d.grid)) return 0;
    auto same = [](const DField& x, const DField& y) {
        return x.ncomp == 1 && y.ncomp == 1 && x.dtype == y.dtype && x.st_t == y.st_t && x.st_z == y.st_z && x.st_y == y.st_y && x.st_x == y.st_x &&
               x.nt == y.nt && x.nz == y.nz && x.ny == y.ny && x.nx == y.nx && x.nslots == y.nslots && x.has_time_interval == y.has_time_interval;
    };
    if (!same(U.d, V.d) || (W && !same(U.d, W->d))) return 0;
    if (V.time != U.time || (W && W->time != U.time)) return 0;
    const DField& f = U.d;
    const size_t esz = f.dtype == PK_F64 ? 8 : 4;
    if (!g.d.has_x || f.st_x != 1 || f.nx < 2 || f.nx != g.d.nx) return 0;
    // an axis the fields extend over must be the grid's axis (then an in-bounds cell index + 1 never needs clipping)
    const bool fy = f.ny >= 2 && f.st_y > 0, fz = f.nz >= 2 && f.st_z > 0;
    if (fy && !(g.d.has_y && f.ny == g.d.ny)) return 0;
    if (fz && !(g.d.has_z && f.nz == g.d.nz)) return 0;
    if (f.has_time_interval && (f.nt < 2 || U.time.front() != 0.0)) return 0;
    const uint64_t lvl_b = (uint64_t)f.st_t * esz;
    const uint64_t last_b = ((uint64_t)f.nz * (uint64_t)f.st_z + (uint64_t)f.ny * (uint64_t)f.st_y + (uint64_t)f.nx) * esz;
    if (lvl_b >= (1ull << 32) || last_b >= (1ull << 32)) return 0;
    // coordinate tables, built once per (grid, field)
    if (ctx->fast_tab_grid != U.d.grid || ctx->fast_tab_field != prm->fU) {
        ctx->fast_tab_grid = U.d.grid;
        ctx->fast_tab_field = prm->fU;
        ctx->fast_tab_ok = false;
        std::vector<double> tab;
        bool ok = true;
        auto push = [&](const double* arr, int n) {
            for (int i = 0; i < n; i++) {
                double r = 0.0;
                if (i + 1 < n) {
                    const double d = arr[i + 1] - arr[i];
                    r = 1.0 / d;
                    if (!(d > 1e-100 && d < 1e100) || !std::isfinite(r)) ok = false;
                }
                tab.push_back(arr[i]);
                tab.push_back(r);
            }
        };
        const int nt = f.has_time_interval ? f.nt : 0;
        const int nz = g.d.has_z ? g.d.nz : 0, ny = g.d.has_y ? g.d.ny : 0, nx = g.d.nx;
        ctx->fast_tab_off[0] = 0;
        push(U.time.data(), nt);
        ctx->fast_tab_off[1] = (int32_t)(tab.size() / 2);
        std::vector<double> tmp;
        auto fetch = [&](const double* dev, int n) -> int32_t {
            tmp.assign((size_t)std::max(n, 0), 0.0);
            if (n > 0) PK_HIP(ctx, hipMemcpy(tmp.data(), dev, sizeof(double) * n, hipMemcpyDeviceToHost));
            return 0;
        };
        if (int32_t rc = fetch(g.d.depth, nz)) return rc;
        push(tmp.data(), nz);
        ctx->fast_tab_off[2] = (int32_t)(tab.size() / 2);
        if (int32_t rc = fetch(g.d.lat, ny)) return rc;
        push(tmp.data(), ny);
        ctx->fast_tab_off[3] = (int32_t)(tab.size() / 2);
        if (int32_t rc = fetch(g.d.lon, nx)) return rc;
        push(tmp.data(), nx);
        ctx->fast_tab_off[4] = (int32_t)(tab.size() / 2);
        if (tab.size() * sizeof(double) > 60 * 1024) ok = false;  // must fit the LDS of a workgroup
        if (ok) {
            if (tab.size() > ctx->fast_tab_cap) {
                PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
                if (ctx->d_fast_tab) PK_HIP(ctx, hipFree(ctx->d_fast_tab));
                PK_HIP(ctx, hipMalloc((void**)&ctx->d_fast_tab, tab.size() * sizeof(double)));
                ctx->fast_tab_cap = tab.size();
            }
            PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
            PK_HIP(ctx, hipMemcpy(ctx->d_fast_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        ctx->fast_tab_ok = ok;
    }
    if (!ctx->fast_tab_ok) return 0;
    const double inv_deg2m = 1.0 / g.d.deg2m;
    if (!(g.d.deg2m > 1e-100 && g.d.deg2m < 1e100) || !std::isfinite(inv_deg2m)) return 0;
    F.grid = U.d.grid;
    F.has_ti = f.has_time_interval;
    F.has_z = g.d.has_z; F.has_y = g.d.has_y; F.has_x = g.d.has_x;
    F.spherical = g.d.spherical;
    F.nt = f.nt;
    F.nslots = f.nslots;
    F.gnz = g.d.nz; F.gny = g.d.ny; F.gnx = g.d.nx;
    uint32_t stride = 1;
    if (g.d.has_x) { F.ex = stride; stride *= (uint32_t)g.d.xdim; }
    if (g.d.has_y) { F.ey = stride; stride *= (uint32_t)g.d.ydim; }
    if (g.d.has_z) { F.ez = stride; }
    F.st_z = (uint32_t)f.st_z;
    F.st_y = (uint32_t)f.st_y;
    F.dyb = fy ? (uint32_t)(f.st_y * esz) : 0u;
    F.dzb = fz ? (uint32_t)(f.st_z * esz) : 0u;
    F.lds_time = ctx->fast_tab_off[0]; F.lds_depth = ctx->fast_tab_off[1]; F.lds_lat = ctx->fast_tab_off[2]; F.lds_lon = ctx->fast_tab_off[3];
    F.lds_n = ctx->fast_tab_off[4];
    F.lvl_b = (int64_t)lvl_b;
    F.U = (const char*)U.d.data; F.V = (const char*)V.d.data; F.W = W ? (const char*)W->d.data : nullptr;
    F.tab = ctx->d_fast_tab;
    F.tlen = f.tlen; F.t0 = f.tfirst; F.t1 = f.tlast;
    F.z0 = g.d.zfirst; F.z1 = g.d.zlast; F.y0 = g.d.yfirst; F.y1 = g.d.ylast; F.x0 = g.d.xfirst; F.x1 = g.d.xlast;
    F.deg2m = g.d.deg2m;
    F.inv_deg2m = inv_deg2m;
    F.ok = 1;
    return 0;
}

// Fold the descriptors of a C-grid velocity and its spherical curvilinear grid into the wave-uniform constants of the fast C-grid
// path (pk_device.h: FastC, pk_fast_cgrid.h).  a.fastc.ok == 0 when a precondition fails (the general program runs): float64 node
// coordinates, spherical mesh, per-cell table present, U / V (/ W) of one shape on the grid's own node counts with staggering offsets
// in {0, 1} (then no staggered index needs clipping), a level below 2^31 elements, every search of the launch guessed.
// The pair copies of FastC::vp for the levels that are resident right now: allocated on first use (no memory for them: the kernels read
// the level rings), (re)packed on the compute stream -- ahead of the launch that is being prepared -- for every pair of adjacent
// committed levels whose copy is missing or older than one of the four uploads it was made from.
static void ensure_velocity_pairs(pk_ctx* ctx, const pk_exec_params* prm, FastC& F) {
    F.vp = nullptr;
    F.vp_slot_b = 0;
    F.vp_hi = 0;
    if (ctx->no_velocity_pairs) return;
    const HostField& U = ctx->fields[prm->fU];
    const HostField& V = ctx->fields[prm->fV];
    const DField& f = U.d;
    if (!f.has_time_interval || f.nt < 2 || f.st_t <= 0) return;
    const size_t esz = f.dtype == PK_F64 ? 8 : 4;
    const int ns = f.nslots;
    const size_t slot_b = (size_t)f.st_t * 8 * esz;
    const size_t need = slot_b * (size_t)ns;
    if (ctx->vp_fU != prm->fU || ctx->vp_fV != prm->fV || ctx->vp_bytes != need || (int)ctx->vp_level.size() != ns) {
        if (ctx->d_vp) {
            (void)hipStreamSynchronize(ctx->compute);
            (void)hipFree(ctx->d_vp);
            ctx->d_vp = nullptr;
        }
        ctx->vp_fU = prm->fU;
        ctx->vp_fV = prm->fV;
        ctx->vp_bytes = need;
        ctx->vp_level.assign(ns, -1);
        ctx->vp_gen.assign((size_t)ns * 4, 0);
        // an opportunistic allocation must not starve what the run needs later (checkpoints, snapshots, further fields): it is made only
        // while it leaves a quarter of the device memory -- and at least its own size again -- free
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need + std::max(need, total_b / 4)) {
            (void)hipGetLastError();
            ctx->d_vp = nullptr;
        } else if (hipMalloc((void**)&ctx->d_vp, need) != hipSuccess) {
            (void)hipGetLastError();
            ctx->d_vp = nullptr;
        }
    }
    if (!ctx->d_vp) return;
    const int64_t cb = F.cb;
    // the resident levels must be ONE run lo..hi of at least two (a ring in steady state, or everything): else the level rings serve
    int lo = -1, hi = -1, count = 0;
    for (int L = 0; L < f.nt; L++) {
        const int sl = L % ns;
        if (U.slot_level[sl] != L || V.slot_level[sl] != L) continue;
        if (lo < 0) lo = L;
        hi = L;
        count++;
    }
    if (lo < 0 || hi == lo || count != hi - lo + 1) return;
    bool timing = false;
    for (int L = lo; L < hi; L++) {
        const int s0 = L % ns, s1 = (L + 1) % ns, ps = L % ns;
        const uint64_t g[4] = {U.slot_gen[s0], U.slot_gen[s1], V.slot_gen[s0], V.slot_gen[s1]};
        uint64_t* have = &ctx->vp_gen[(size_t)ps * 4];
        if (ctx->vp_level[ps] == L && have[0] == g[0] && have[1] == g[1] && have[2] == g[2] && have[3] == g[3]) continue;
        const int64_t n = f.st_t;
        const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 256 * 64);
        char* out = ctx->d_vp + (size_t)ps * slot_b;
        if (!timing) {
            (void)hipEventRecord(ctx->ev_p0, ctx->compute);
            timing = true;
        }
        const int64_t o0 = (int64_t)s0 * F.lvl_b, o1 = (int64_t)s1 * F.lvl_b;
        if (esz == 8)
            hipLaunchKernelGGL((cg_pack_pairs_kernel<double>), dim3(grid), dim3(256), 0, ctx->compute, F.U, F.V, F.dU0, F.dU1, F.dV0, F.dV1, cb, o0, o1, n,
                               (int32_t)f.ny, (int32_t)f.nx, (double*)out);
        else
            hipLaunchKernelGGL((cg_pack_pairs_kernel<float>), dim3(grid), dim3(256), 0, ctx->compute, F.U, F.V, F.dU0, F.dU1, F.dV0, F.dV1, cb, o0, o1, n,
                               (int32_t)f.ny, (int32_t)f.nx, (float*)out);
        if (hipGetLastError() != hipSuccess) return;  // (the level rings serve)
        ctx->vp_level[ps] = L;
        for (int k = 0; k < 4; k++) have[k] = g[k];
        ctx->fl_packs++;
    }
    if (timing) (void)hipEventRecord(ctx->ev_p1, ctx->compute);
    F.vp_hi = hi;
    // every pair inside the window a launch may touch is packed now: the resident window is made of committed levels only
    F.vp = ctx->d_vp;
    F.vp_slot_b = (int64_t)slot_b;
}

static int32_t fill_fastc(pk_ctx* ctx, const pk_exec_params* prm, KArgs& a, bool want_w, size_t& lds_bytes, bool rk45 = false, bool m1 = false) {
    FastC& F = a.fastc;
    memset(&F, 0, sizeof(F));
    // the RK4 kernels reset dt every iteration (kernel.py:225-226); in RK45 mode (fieldset.RK45_tol present) dt follows next_dt, which
    // only the RK45 kernel implements -- and AdvectionRK45 itself always runs in that mode (Kernel.check_fieldsets_in_kernels)
    if (ctx->no_fast_cgrid || prm->interp_uv != 1 || (prm->rk45_mode != 0) != rk45) return 0;
    if (rk45 && !ctx->dev.next_dt) return 0;
    if (prm->reset_state && !prm->have_guess0) return 0;  // an unguessed first search returns float32 (xsi, eta) ARRAYS (GPos::w32)
    const HostField& U = ctx->fields[prm->fU];
    const HostField& V = ctx->fields[prm->fV];
    const HostField* W = (want_w && prm->fW >= 0) ? &ctx->fields[prm->fW] : nullptr;
    if (want_w && !W) return 0;
    const int gid = U.d.grid;
    const HostGrid& g = ctx->grids[gid];
    if (g.d.kind != 1 || !g.d.spherical || g.d.lon_f32 || g.d.lat_f32 || g.d.depth_f32 || !g.d.cell_tab) return 0;
    if (V.d.grid != gid || (W && W->d.grid != gid)) return 0;
    auto same = [](const DField& x, const DField& y) {
        return x.ncomp == y.ncomp && x.dtype == y.dtype && x.st_t == y.st_t && x.st_z == y.st_z && x.st_y == y.st_y && x.st_x == y.st_x &&
               x.nt == y.nt && x.nz == y.nz && x.ny == y.ny && x.nx == y.nx && x.nslots == y.nslots && x.has_time_interval == y.has_time_interval;
    };
    if (!same(U.d, V.d) || (W && !same(U.d, W->d))) return 0;
    if (V.time != U.time || (W && W->time != U.time)) return 0;
    const DField& f = U.d;
    const int64_t esz = f.dtype == PK_F64 ? 8 : 4;
    if (!g.d.has_x || !g.d.has_y || f.st_x != 1 || f.nx != g.d.nx || f.ny != g.d.ny || g.d.nx < 2 || g.d.ny < 2) return 0;
    for (int o : {g.d.off_x, g.d.off_y, g.d.off_z})
        if (o != 0 && o != 1) return 0;
    if (g.d.has_z) {
        if (g.d.nz < 2 || f.nz < g.d.nz + (W ? g.d.off_z : 0) || g.d.nz >= 4096) return 0;
    } else if (W) {
        return 0;
    }
    if (f.has_time_interval && (f.nt < 2 || U.time.front() != 0.0)) return 0;
    if (f.nt >= (1 << 18)) return 0;
    if ((int64_t)f.st_t >= (1ll << 31)) return 0;
    // `ei` must not wrap (the guess of the next search is the cell itself)
    if ((int64_t)std::max(g.d.xdim, 1) * std::max(g.d.ydim, 1) * std::max(g.d.zdim, 1) >= (1ll << 31)) return 0;
    if (!(g.d.deg2m > 1e-100 && g.d.deg2m < 1e100)) return 0;
    // coordinate tables of time | depth, built once per (grid, field)
    if (ctx->cg_tab_grid != gid || ctx->cg_tab_field != prm->fU) {
        ctx->cg_tab_grid = gid;
        ctx->cg_tab_field = prm->fU;
        ctx->cg_tab_ok = false;
        std::vector<double> tab;
        bool ok = true;
        auto push = [&](const double* arr, int n) {
            for (int i = 0; i < n; i++) {
                double r = 0.0;
                if (i + 1 < n) {
                    const double d = arr[i + 1] - arr[i];
                    r = 1.0 / d;
                    if (!(d > 1e-100 && d < 1e100) || !std::isfinite(r)) ok = false;
                }
                tab.push_back(arr[i]);
                tab.push_back(r);
            }
        };
        const int nt = f.has_time_interval ? f.nt : 0;
        const int nz = g.d.has_z ? g.d.nz : 0;
        ctx->cg_tab_off[0] = 0;
        push(U.time.data(), nt);
        ctx->cg_tab_off[1] = (int32_t)(tab.size() / 2);
        std::vector<double> tmp((size_t)std::max(nz, 0), 0.0);
        if (nz > 0) PK_HIP(ctx, hipMemcpy(tmp.data(), g.d.depth, sizeof(double) * nz, hipMemcpyDeviceToHost));
        push(tmp.data(), nz);
        ctx->cg_tab_off[2] = (int32_t)(tab.size() / 2);
        if (tab.size() * sizeof(double) > 24 * 1024) ok = false;  // next to the per-lane slots in the LDS of a one-wavefront workgroup
        if (tab.empty()) tab.assign(2, 0.0);
        if (ok) {
            PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
            if (tab.size() > ctx->cg_tab_cap) {
                if (ctx->d_cg_tab) PK_HIP(ctx, hipFree(ctx->d_cg_tab));
                PK_HIP(ctx, hipMalloc((void**)&ctx->d_cg_tab, tab.size() * sizeof(double)));
                ctx->cg_tab_cap = tab.size();
            }
            PK_HIP(ctx, hipMemcpy(ctx->d_cg_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        ctx->cg_tab_ok = ok;
    }
    if (!ctx->cg_tab_ok) return 0;
    // per-cell records, built once per grid (optional: without the memory for them the general program runs)
    if (ctx->ct2_grid != gid) {
        if (ctx->d_ct2) {
            PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
            (void)hipFree(ctx->d_ct2);
            ctx->d_ct2 = nullptr;
        }
        ctx->ct2_grid = gid;
        const int64_t ncell = (int64_t)(g.d.ny - 1) * g.d.nx;
        if (hipMalloc((void**)&ctx->d_ct2, (size_t)ncell * CT2_STRIDE * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            ctx->d_ct2 = nullptr;
        } else {
            int* d_wide = nullptr;
            int h_wide = 1;
            PK_HIP(ctx, hipMalloc((void**)&d_wide, sizeof(int)));
            PK_HIP(ctx, hipMemsetAsync(d_wide, 0, sizeof(int), ctx->compute));
            hipLaunchKernelGGL(cell_table2_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, ctx->compute, g.d, ctx->d_ct2, d_wide);
            PK_HIP(ctx, hipGetLastError());
            PK_HIP(ctx, hipMemcpyAsync(&h_wide, d_wide, sizeof(int), hipMemcpyDeviceToHost, ctx->compute));
            PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
            (void)hipFree(d_wide);
            ctx->ct2_near = h_wide ? 0 : 1;
        }
    }
    if (!ctx->d_ct2) return 0;
    F.grid = gid;
    F.has_ti = f.has_time_interval;
    F.has_z = g.d.has_z;
    F.walk_ok = g.d.walk_ok;
    F.near_edges = ctx->ct2_near;
    F.nt = f.nt;
    F.nslots = f.nslots;
    F.gnz = g.d.nz; F.gny = g.d.ny; F.gnx = g.d.nx;
    uint32_t stride = 1;
    if (g.d.has_x) { F.ex = stride; stride *= (uint32_t)g.d.xdim; }
    if (g.d.has_y) { F.ey = stride; stride *= (uint32_t)g.d.ydim; }
    if (g.d.has_z) { F.ez = stride; }
    F.st_z = (int32_t)f.st_z;
    F.st_y = (int32_t)f.st_y;
    const int64_t cb = (int64_t)f.ncomp * esz;
    F.cb = (int32_t)cb;
    F.lvl_b = (int64_t)f.st_t * cb;
    const int64_t oz = g.d.has_z ? g.d.off_z : 0;
    F.dU0 = (int64_t)g.d.off_y * f.st_y * cb + (int64_t)U.d.comp * esz;
    F.dU1 = F.dU0 + cb;
    F.dV0 = (int64_t)g.d.off_x * cb + (int64_t)V.d.comp * esz;
    F.dV1 = F.dV0 + (int64_t)f.st_y * cb;
    if (W) {
        F.dW0 = (oz * f.st_z + (int64_t)g.d.off_y * f.st_y + g.d.off_x) * cb + (int64_t)W->d.comp * esz;
        F.dW1 = F.dW0 + (int64_t)f.st_z * cb;
    }
    F.U = (const char*)U.d.data; F.V = (const char*)V.d.data; F.W = W ? (const char*)W->d.data : nullptr;
    F.ct2 = ctx->d_ct2;
    F.tab = ctx->d_cg_tab;
    if (m1) {  // AdvectionDiffusionM1: Kh_zonal / Kh_meridional as XLinear fields on the nodes of the same grid
        const int ids[2] = {prm->fKh_zonal, prm->fKh_meridional};
        for (int k = 0; k < 2; k++) {
            if (ids[k] < 0 || ids[k] >= (int)ctx->fields.size()) return 0;
            const HostField& K = ctx->fields[ids[k]];
            const DField& kd = K.d;
            if (kd.grid != gid || kd.dtype != f.dtype || kd.ncomp != 1 || kd.is_const != 0 || kd.st_x != 1 || kd.nx != g.d.nx || kd.ny != g.d.ny) return 0;
            if (!(kd.nz == 1 || kd.nz == g.d.nz)) return 0;
            if (kd.has_time_interval && (K.time != U.time || kd.nt != f.nt)) return 0;
            if (!kd.has_time_interval && kd.nt != 1) return 0;
            if ((int64_t)kd.st_t * esz >= (1ll << 31)) return 0;
            F.kh[k] = (const char*)kd.data;
            F.kh_st[k] = (int32_t)(kd.st_t * esz);
            F.kh_sz[k] = (int32_t)(kd.st_z * esz);
            F.kh_sy[k] = (int32_t)(kd.st_y * esz);
            F.kh_nt[k] = kd.nt; F.kh_nz[k] = kd.nz; F.kh_ny[k] = kd.ny; F.kh_nx[k] = kd.nx;
            F.kh_has_ti[k] = kd.has_time_interval;
            F.kh_nslots[k] = kd.nslots;
        }
    }
    F.lds_time = ctx->cg_tab_off[0]; F.lds_depth = ctx->cg_tab_off[1]; F.lds_n = ctx->cg_tab_off[2];
    F.lds_rec = 2 * F.lds_n;
    const int cm = m1 ? CG_CACHE_M1 : (rk45 ? CG_CACHE_RK45 : CG_CACHE_RK4);  // which parts of a lane's cell cache the kernel keeps in registers
    F.lds_fv = F.lds_rec + fc_rec_rows(cm) * FC_LANES;
    lds_bytes = (size_t)F.lds_fv * sizeof(double) + (size_t)fc_fv_lds(cm) * FC_LANES * (size_t)esz;
    F.tlen = f.tlen; F.t0 = f.tfirst; F.t1 = f.tlast;
    F.z0 = g.d.zfirst; F.z1 = g.d.zlast;
    F.deg2m = g.d.deg2m;
    if (!W) ensure_velocity_pairs(ctx, prm, F);
    F.ok = 1;
    return 0;
}

// A kernel list WITH user kernels may still run in the dedicated A-grid kernel (the module carries an instantiation with the user kernels
// as side kernels, pk_kernels.h: side_kernel): exactly one AdvectionRK4 / AdvectionRK4_3D anywhere in the list, everything else a
// sampling-free recovery kernel or a user kernel, and a module whose user kernels sample no field.  -1: no; 0 / 1: yes, 2-D / 3-D.
static int user_fast_shape(const pk_exec_params* prm, int32_t user_flags) {
    if (!(user_flags & PK_USER_RIDE) || prm->body_only) return -1;
    int nadv = 0, d3 = 0;
    for (int k = 0; k < prm->nk; k++) {
        const int id = prm->kernels[k];
        if (id == PK_KERNEL_ADVECTION_RK4 || id == PK_KERNEL_ADVECTION_RK4_3D) {
            nadv++;
            d3 = id == PK_KERNEL_ADVECTION_RK4_3D;
        } else if (!(id == PK_KERNEL_DELETE_ON_ERROR || id == PK_KERNEL_DELETE_OUT_OF_BOUNDS || (id >= PK_KERNEL_USER0 && id < PK_KERNEL_USER0 + PK_MAX_USER_KERNELS))) {
            return -1;
        }
    }
    if (nadv != 1) return -1;
    if ((user_flags & PK_USER_SAMPLES_UVW) && !d3) return -1;  // a 2-D host kernel carries no W
    return d3;
}
// the scalar fields a riding module samples must look exactly like U to the dedicated A-grid kernel: same grid, dtype, layout, time axis,
// ring, XLinear -- then only the base pointer differs (FastA::S)
static bool fill_fast_scalars(pk_ctx* ctx, const pk_exec_params* prm, FastA& F, int nsample, const int32_t* fids) {
    if (nsample < 0 || nsample > 4) return false;
    const HostField& U = ctx->fields[prm->fU];
    F.ns = 0;
    for (int k = 0; k < nsample; k++) {
        const int fid = fids[k];
        if (fid < 0 || fid >= (int)ctx->fields.size()) return false;
        const HostField& S = ctx->fields[fid];
        const DField &x = S.d, &y = U.d;
        const bool same = x.grid == y.grid && x.ncomp == 1 && x.dtype == y.dtype && x.st_t == y.st_t && x.st_z == y.st_z && x.st_y == y.st_y &&
                          x.st_x == y.st_x && x.nt == y.nt && x.nz == y.nz && x.ny == y.ny && x.nx == y.nx && x.nslots == y.nslots &&
                          x.has_time_interval == y.has_time_interval && x.is_const == 0 && S.time == U.time;
        if (!same) return false;
        F.sfid[F.ns] = fid;
        F.S[F.ns] = (const char*)x.data;
        F.ns++;
    }
    return true;
}

int32_t pk_execute_begin(pk_ctx* ctx, const pk_exec_params* prm) {
    if (!ctx || !prm) return -2;
    if (!ctx->bound) return ctx->fail("no particles bound");
    if (ctx->in_flight) return ctx->fail("pk_execute_begin: a launch is already in flight (call pk_execute_end)");
    if (prm->nk < 1 || prm->nk > PK_MAX_KERNELS) return ctx->fail("params.nk out of range");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    ctx->fl_clock_probe = false;  // only the branch that launches a probe sets it: no launch path reports the clock of an older launch
    KArgs a;
    size_t lds_bytes = 0;
    int use_lds = 0;
    int32_t rc = fill_args(ctx, prm, a, lds_bytes, use_lds);
    if (rc) return rc;
    bool need_kh = false, has_user = false;
    for (int k = 0; k < prm->nk; k++) {
        const int id = prm->kernels[k];
        if (id == PK_KERNEL_ADVECTIONDIFFUSION_M1 || id == PK_KERNEL_ADVECTIONDIFFUSION_EM || id == PK_KERNEL_DIFFUSION_UNIFORM_KH)
            need_kh = true;
        if (id == PK_KERNEL_ADVECTION_RK45 && !ctx->dev.next_dt) return ctx->fail("AdvectionRK45 needs the next_dt column");
        if (id >= PK_KERNEL_USER0 && id < PK_KERNEL_USER0 + PK_MAX_USER_KERNELS) {
            has_user = true;
            if (!ctx->user_launch) return ctx->fail("a user kernel id without a registered program (pk_set_user_program)");
            if (prm->body_only) return ctx->fail("user kernels do not run in body_only launches");
        }
        if ((id == PK_KERNEL_ADVECTION_RK4_3D || id == PK_KERNEL_ADVECTION_RK2_3D) && prm->fW < 0)
            return ctx->fail("3-D advection needs the W field");
        if (id == PK_KERNEL_SAMPLE_FIELD) {
            const int sf = prm->sample_field[k];
            const bool vec = sf == PK_SAMPLE_UV || sf == PK_SAMPLE_UVW;
            if (!vec && (sf < 0 || sf >= (int)ctx->fields.size())) return ctx->fail("PK_KERNEL_SAMPLE_FIELD: params.sample_field names no field");
            if (sf == PK_SAMPLE_UVW && prm->fW < 0) return ctx->fail("PK_KERNEL_SAMPLE_FIELD: sampling UVW needs the W field");
            const int nv = vec ? (sf == PK_SAMPLE_UVW ? 3 : 2) : 1;
            int stored = 0;
            for (int j = 0; j < nv; j++) {
                const int v = vec ? ((prm->sample_var[k] >> (8 * j)) & 0xFF) : prm->sample_var[k];
                if (vec && v == PK_SAMPLE_DISCARD) continue;
                if (v < 0 || v >= ctx->host.n_extra) return ctx->fail("PK_KERNEL_SAMPLE_FIELD: params.sample_var names no extra particle column");
                stored++;
            }
            if (!stored) return ctx->fail("PK_KERNEL_SAMPLE_FIELD: every component is discarded");
        }
    }
    if (need_kh && (prm->fKh_zonal < 0 || prm->fKh_meridional < 0)) return ctx->fail("diffusion kernels need Kh_zonal/Kh_meridional");
    const HostField& U = ctx->fields[prm->fU];
    const int field_f32 = U.d.dtype == PK_F32;
    for (int f : {prm->fV, prm->fW})
        if (f >= 0 && ctx->fields[f].d.dtype != U.d.dtype) return ctx->fail("U, V, W must share one dtype");
    const int curv = ctx->grids[a.main_grid].d.kind == 1;
    const int64_t n = ctx->dev.n;
    if (prm->twe_n < 0 || prm->twe_n > PK_MAX_TWE || (prm->twe_n > 0 && !prm->twe_key)) return ctx->fail("params.twe_n / twe_key out of range");
    for (int k = 1; k < prm->twe_n; k++)
        if (!(prm->twe_key[k - 1] < prm->twe_key[k])) return ctx->fail("params.twe_key must be ascending");
    DCounters& counters0 = ctx->h_counters0;
    counters0 = DCounters{0ull, 0ull, 0ull, 0xFFFFFFFFu, 0u, ~0ull, 0u, 0u};
    PK_HIP(ctx, hipMemcpyAsync(ctx->d_counters, &counters0, sizeof(DCounters), hipMemcpyHostToDevice, ctx->compute));
    PK_HIP(ctx, hipMemsetAsync(ctx->d_twe_found, 0, sizeof(unsigned long long) * TWE_FOUND_SLOTS, ctx->compute));
    if (prm->twe_n > 0) PK_HIP(ctx, hipMemsetAsync(ctx->d_twe_hit, 0, sizeof(unsigned int) * prm->twe_n, ctx->compute));
    ctx->fl_twe_n = prm->twe_n;
    if (prm->twe_key != ctx->rerun_keys.data()) ctx->rerun_keys.assign(prm->twe_key, prm->twe_key + prm->twe_n);  // (kept for pk_execute_rerun)
    if (prm->twe_n > 0)
        PK_HIP(ctx, hipMemcpyAsync(ctx->d_twe, ctx->rerun_keys.data(), sizeof(int64_t) * prm->twe_n, hipMemcpyHostToDevice, ctx->compute));
    PK_HIP(ctx, hipMemsetAsync(ctx->d_summary, 0, sizeof(unsigned long long) * PK_NUM_STATE_CODES, ctx->compute));
    const unsigned long long init_mm[2] = {~0ull, 0ull};
    PK_HIP(ctx, hipMemcpyAsync(ctx->d_summary + PK_NUM_STATE_CODES, init_mm, sizeof(init_mm), hipMemcpyHostToDevice, ctx->compute));
    int launches = 0;
    bool sorted = false;
    if (n > 0) {
        const dim3 grid((unsigned)((n + 255) / 256));
        int prog = PROG_GENERIC;
        bool rest_policy = true;  // entries after the first are the sampling-free recovery kernels
        for (int k = 1; k < prm->nk; k++)
            rest_policy = rest_policy && (prm->kernels[k] == PK_KERNEL_DELETE_ON_ERROR || prm->kernels[k] == PK_KERNEL_DELETE_OUT_OF_BOUNDS);
        if (rest_policy && use_lds) {
            if (prm->kernels[0] == PK_KERNEL_ADVECTION_RK4) prog = PROG_RK4;
            if (prm->kernels[0] == PK_KERNEL_ADVECTION_RK4_3D) prog = PROG_RK4_3D;
            if (prm->kernels[0] == PK_KERNEL_ADVECTION_RK45 && !ctx->no_special) prog = PROG_RK45;
            if (prm->kernels[0] == PK_KERNEL_ADVECTIONDIFFUSION_M1 && !ctx->no_special) prog = PROG_M1;
        }
        if (prm->body_only) prog = PROG_GENERIC;  // only the kernel-list interpreter knows the mode
        if (ctx_is_typed(ctx)) prog = PROG_TYPED;
        if (has_user && prog != PROG_GENERIC) return ctx->fail("user kernels run in the plain kernel-list interpreter only (float32 coordinate arrays are not supported)");
        bool fast_a = false, fast_c = false;  // a.fast / a.fastc share storage: at most one is filled
        size_t cgrid_lds = 0;
        // A launch that knows samples which fail call-wide (pk_exec_params.twe_key: the repeat of a call in which some particle left a
        // field's time interval) runs the general programs: only they test the list (the dedicated kernels report such samples and pay
        // nothing else for it); same results otherwise, which the parity tests hold them to at rtol 0.  (Round 6 built the list test into the
        // dedicated kernels -- one scalar bit test per sample when nothing is listed -- held it bit-identical and measured it: +2.4 % on the
        // headline kernel, +0.6 % / +1 % on RK45 / M1 through the register allocation alone (profiles/r06e_summary.txt), for repeats that only
        // a run past the last time level makes.  Not kept.)
        const bool listed = prm->twe_n > 0;
        const int ufast = (has_user && use_lds && !listed) ? user_fast_shape(prm, ctx->user_flags) : -1;
        const bool user_samples = ctx->user_nsample > 0 || (ctx->user_flags & (PK_USER_SAMPLES_UV | PK_USER_SAMPLES_UVW));
        if (ufast >= 0 && !curv) {
            rc = fill_fast(ctx, prm, a, ufast == 1);
            if (rc) return rc;
            if (a.fast.ok && !fill_fast_scalars(ctx, prm, a.fast, ctx->user_nsample, ctx->user_sample_fid)) a.fast.ok = 0;
            fast_a = a.fast.ok != 0;
            if (fast_a) prog = ufast ? PROG_RK4_3D : PROG_RK4;
        } else if (ufast >= 0 && curv && !user_samples) {
            rc = fill_fastc(ctx, prm, a, ufast == 1, cgrid_lds);
            if (rc) return rc;
            fast_c = a.fastc.ok != 0;
            if (fast_c) prog = ufast ? PROG_RK4_3D : PROG_RK4;
        } else if (listed) {
            // (general program)
        } else if ((prog == PROG_RK4 || prog == PROG_RK4_3D) && !curv) {
            rc = fill_fast(ctx, prm, a, prog == PROG_RK4_3D);
            if (rc) return rc;
            fast_a = a.fast.ok != 0;
        } else if ((prog == PROG_RK4 || prog == PROG_RK4_3D || prog == PROG_RK45 || prog == PROG_M1) && curv) {
            rc = fill_fastc(ctx, prm, a, prog == PROG_RK4_3D, cgrid_lds, prog == PROG_RK45, prog == PROG_M1);
            if (rc) return rc;
            fast_c = a.fastc.ok != 0;
        }
        if (prm->sort_by_cell && !prm->body_only) {
            PK_HIP(ctx, hipEventRecord(ctx->ev2, ctx->compute));
            // curvilinear sort order (measured on the NEMO-size grid): depth-major for 3-D advection (+4 %), horizontal-major
            // (water columns share node-table lines) for 2-D kernels (RK45 +18 %, M1 +28 %); PK_SORT_HORIZONTAL overrides
            int horizontal = 1;
            for (int k = 0; k < prm->nk; k++)
                if (prm->kernels[k] == PK_KERNEL_ADVECTION_RK4_3D || prm->kernels[k] == PK_KERNEL_ADVECTION_RK2_3D) horizontal = 0;
            if (ctx->sort_horizontal_major >= 0) horizontal = ctx->sort_horizontal_major;
            rc = sort_particles(ctx, a.main_grid, horizontal);
            if (rc) return rc;
            a.p = ctx->dev;  // the column pointers were swapped
            sorted = true;
        }
        // the launch reads `dev` and writes the second column set (pk_device.h: DPOut), which then becomes `dev`: the state before
        // the launch survives in `alt` at no cost, for pk_execute_rerun
        rc = ensure_alt(ctx);
        if (rc) return rc;
        a.p = ctx->dev;
        a.po = DPOut{ctx->alt.t, ctx->alt.z, ctx->alt.y, ctx->alt.x, ctx->alt.dz, ctx->alt.dy, ctx->alt.dx, ctx->alt.dt, ctx->alt.next_dt,
                     ctx->alt.state, ctx->alt.ei, ctx->alt.iter};
        if (!prm->reset_state || prm->body_only) {
            // a continued (paused) call or a masked body launch: some rows are not stepped -- they keep their values by this
            // device-to-device copy of the columns a launch writes (the kernels themselves carry no copy code: pk_kernels.h)
            const size_t ss = spatial_size(ctx);
            const size_t nn = (size_t)n;
            const struct { void* dst; const void* src; size_t bytes; } cp[] = {
                {ctx->alt.t, ctx->dev.t, nn * 8}, {ctx->alt.z, ctx->dev.z, nn * ss}, {ctx->alt.y, ctx->dev.y, nn * ss}, {ctx->alt.x, ctx->dev.x, nn * ss},
                {ctx->alt.dz, ctx->dev.dz, nn * ss}, {ctx->alt.dy, ctx->dev.dy, nn * ss}, {ctx->alt.dx, ctx->dev.dx, nn * ss}, {ctx->alt.dt, ctx->dev.dt, nn * 8},
                {ctx->alt.next_dt, ctx->dev.next_dt, nn * 8}, {ctx->alt.state, ctx->dev.state, nn * 4},
                {ctx->alt.ei, ctx->dev.ei, nn * 4 * (size_t)ctx->host.ngrids}, {ctx->alt.iter, ctx->dev.iter, nn * 4}};
            for (const auto& c : cp)
                if (c.dst && c.src) PK_HIP(ctx, hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDeviceToDevice, ctx->compute));
        }
        if (ctx->clock_probe) {  // the probe may start once everything queued before the advection kernel is done
            if (!ctx->probe) {
                PK_HIP(ctx, hipStreamCreateWithFlags(&ctx->probe, hipStreamNonBlocking));
                PK_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_probe, hipEventDisableTiming));
            }
            PK_HIP(ctx, hipMemsetAsync(ctx->d_clk, 0, sizeof(unsigned long long) * 32, ctx->compute));
            PK_HIP(ctx, hipEventRecord(ctx->ev_probe, ctx->compute));
            PK_HIP(ctx, hipStreamWaitEvent(ctx->probe, ctx->ev_probe, 0));
        }
        PK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->compute));
        // the A-grid kernel's LDS: the coordinate tables and, behind them (2-D kernels), 64 bytes per lane of corner-block cache
        size_t fast_lds = fast_a ? (size_t)a.fast.lds_n * 2 * sizeof(double) : 0;
        if (fast_a) {
            const bool blk = prog != PROG_RK4_3D && fast_lds + (size_t)FAST_BLK_BYTES <= 64 * 1024 && !getenv("PK_NO_BLOCK_CACHE");
            a.fast.lds_blk = blk ? a.fast.lds_n : 0;
            if (blk) fast_lds += (size_t)FAST_BLK_BYTES;
        }
        const int pf32 = ctx->dev.spatial_f32;
        if (has_user && fast_a) ctx->user_launch(&a, prog == PROG_RK4_3D ? 2 : 1, field_f32 * 2 + pf32, 1, (uint64_t)fast_lds, (void*)ctx->compute);
        else if (has_user && fast_c) ctx->user_launch(&a, prog == PROG_RK4_3D ? 4 : 3, field_f32 * 2 + pf32, 1, (uint64_t)cgrid_lds, (void*)ctx->compute);
        else if (fast_a && prog == PROG_RK4) launch_fast<PROG_RK4>(field_f32, pf32, a, grid, fast_lds, ctx->compute);
        else if (fast_a && prog == PROG_RK4_3D) launch_fast<PROG_RK4_3D>(field_f32, pf32, a, grid, fast_lds, ctx->compute);
        else if (fast_c && prog == PROG_RK45) launch_cgrid_rk45(field_f32, pf32, a, n, cgrid_lds, ctx->compute);
        else if (fast_c && prog == PROG_M1) launch_cgrid_m1(field_f32, pf32, a, n, cgrid_lds, ctx->compute);
        else if (fast_c) launch_cgrid(field_f32, pf32, prog == PROG_RK4_3D, a, n, cgrid_lds, ctx->compute);
        else switch (prog) {
            case PROG_RK4: launch_program<PROG_RK4>(field_f32, curv, prm->interp_uv, use_lds, a, grid, lds_bytes, ctx->compute); break;
            case PROG_RK4_3D: launch_program<PROG_RK4_3D>(field_f32, curv, prm->interp_uv, use_lds, a, grid, lds_bytes, ctx->compute); break;
            case PROG_RK45: launch_program<PROG_RK45>(field_f32, curv, prm->interp_uv, use_lds, a, grid, lds_bytes, ctx->compute); break;
            case PROG_M1: launch_program<PROG_M1>(field_f32, curv, prm->interp_uv, use_lds, a, grid, lds_bytes, ctx->compute); break;
            case PROG_TYPED: launch_program<PROG_TYPED>(field_f32, curv, prm->interp_uv, use_lds, a, grid, lds_bytes, ctx->compute); break;
            default:
                if (has_user) {
                    const int ik = prm->interp_uv >= 2 ? 2 : prm->interp_uv;
                    ctx->user_launch(&a, 0, (field_f32 ? 6 : 0) + (curv ? 3 : 0) + ik, use_lds, (uint64_t)lds_bytes, (void*)ctx->compute);
                } else {
                    launch_program<PROG_GENERIC>(field_f32, curv, prm->interp_uv, use_lds, a, grid, lds_bytes, ctx->compute);
                }
                break;
        }
        PK_HIP(ctx, hipGetLastError());
        PK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->compute));
        if (ctx->clock_probe) {  // beside the kernel that was just queued: spins for clock_probe microseconds (1 -> 1000) of the 100 MHz counter
            const unsigned long long us = ctx->clock_probe == 1 ? 1000ull : (unsigned long long)std::max(ctx->clock_probe, 20);
            hipLaunchKernelGGL(clock_probe_kernel, dim3(16), dim3(64), 0, ctx->probe, ctx->d_clk, us * 100ull);
            PK_HIP(ctx, hipMemcpyAsync(ctx->h_clk, ctx->d_clk, sizeof(unsigned long long) * 32, hipMemcpyDeviceToHost, ctx->probe));
        }
        ctx->fl_clock_probe = ctx->clock_probe != 0;
        launches = 1;
        ctx->fl_program = fast_a ? 100 : (fast_c ? 101 : prog);
        swap_launch_outputs(ctx);
        ctx->rerun_valid = true;
        ctx->rerun_prm = *prm;
        ctx->rerun_prm.twe_key = nullptr;  // (the caller's array may be gone: the keys live in ctx->rerun_keys)
        const unsigned sgrid = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
        hipLaunchKernelGGL(summarize_kernel, dim3(sgrid), dim3(256), 0, ctx->compute, ctx->dev.state, ctx->dev.t, n, ctx->d_summary);
        PK_HIP(ctx, hipGetLastError());
    }
    PK_HIP(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, sizeof(DCounters), hipMemcpyDeviceToHost, ctx->compute));
    PK_HIP(ctx, hipMemcpyAsync(ctx->h_summary, ctx->d_summary, sizeof(unsigned long long) * (PK_NUM_STATE_CODES + 2), hipMemcpyDeviceToHost, ctx->compute));
    ctx->in_flight = true;
    ctx->fl_launches = launches;
    ctx->fl_sorted = sorted;
    ctx->fl_n = n;
    return 0;
}

int32_t pk_execute_end(pk_ctx* ctx, pk_exec_stats* stats) {
    if (!ctx) return -2;
    if (!ctx->in_flight) return ctx->fail("pk_execute_end: no launch in flight");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    ctx->in_flight = false;
    PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
    if (ctx->fl_clock_probe && ctx->probe) PK_HIP(ctx, hipStreamSynchronize(ctx->probe));
    {
        // (whether or not the caller wants statistics -- ADVICE r5: pk_execute_twe_report after pk_execute_end(ctx, NULL) returned the keys of
        // an older launch)
        const DCounters& hc = *ctx->h_counters;
        // the cold path of the call-wide time error: every unlisted failing sample the general programs reported, and which listed samples a lane
        // justified (pk_execute_twe_report hands them out) -- copied only when there is something to report
        ctx->twe_found_host.clear();
        ctx->twe_hit_host.assign((size_t)ctx->fl_twe_n, 0);
        ctx->twe_overflow_host = hc.twe_overflow != 0;
        if (ctx->fl_launches && hc.twe_key != ~0ull) {
            std::vector<unsigned long long> slots(TWE_FOUND_SLOTS);
            PK_HIP(ctx, hipMemcpy(slots.data(), ctx->d_twe_found, sizeof(unsigned long long) * TWE_FOUND_SLOTS, hipMemcpyDeviceToHost));
            for (unsigned long long k : slots)
                if (k) ctx->twe_found_host.push_back((int64_t)k);
            std::sort(ctx->twe_found_host.begin(), ctx->twe_found_host.end());
        }
        if (ctx->fl_launches && ctx->fl_twe_n > 0) {
            std::vector<unsigned int> hit((size_t)ctx->fl_twe_n);
            PK_HIP(ctx, hipMemcpy(hit.data(), ctx->d_twe_hit, sizeof(unsigned int) * ctx->fl_twe_n, hipMemcpyDeviceToHost));
            for (int k = 0; k < ctx->fl_twe_n; k++) ctx->twe_hit_host[k] = hit[k] ? 1 : 0;
        }
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        const DCounters& hc = *ctx->h_counters;
        const unsigned long long* hs = ctx->h_summary;
        stats->steps = (int64_t)hc.steps;
        stats->attempts = (int64_t)hc.attempts;
        stats->paused = (int64_t)hc.paused;
        for (int k = 0; k < PK_NUM_STATE_CODES; k++) stats->state_counts[k] = (int64_t)hs[k];
        auto unorder = [](unsigned long long o) {
            unsigned long long b = (o & 0x8000000000000000ull) ? (o & 0x7fffffffffffffffull) : ~o;
            double v;
            memcpy(&v, &b, 8);
            return v;
        };
        const bool any_live = ctx->fl_n > 0 && hs[PK_EVALUATE] > 0;
        stats->t_min_live = any_live ? unorder(hs[PK_NUM_STATE_CODES]) : NAN;
        stats->t_max_live = any_live ? unorder(hs[PK_NUM_STATE_CODES + 1]) : NAN;
        float ms = 0.f;
        if (ctx->fl_launches) PK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        stats->kernel_ms = ms;
        float sms = 0.f;
        if (ctx->fl_sorted) PK_HIP(ctx, hipEventElapsedTime(&sms, ctx->ev2, ctx->ev0));
        stats->sort_ms = sms;
        stats->launches = ctx->fl_launches;
        stats->program = ctx->fl_launches ? ctx->fl_program : 0;
        stats->first_error_iter = (ctx->fl_launches && hc.err_iter != 0xFFFFFFFFu) ? (int64_t)hc.err_iter : 0;
        stats->first_time_error_key = (ctx->fl_launches && hc.twe_key != ~0ull) ? (int64_t)hc.twe_key : 0;
        // shader clock beside the advection kernel: cycles per 100 MHz tick of every probe wavefront (the median of the 16)
        double sclk = 0.0;
        int nx = 0;
        if (ctx->fl_launches && ctx->fl_clock_probe) {
            std::vector<double> v;
            for (int k = 0; k < 16; k++) {
                const unsigned long long dc = ctx->h_clk[k * 2], dr = ctx->h_clk[k * 2 + 1];
                if (dc && dr) v.push_back((double)dc / (double)dr * 100.0);
            }
            if (!v.empty()) {
                std::sort(v.begin(), v.end());
                sclk = v[v.size() / 2];
                nx = 1;
            }
        }
        stats->sclk_mhz = nx ? sclk / nx : 0.0;
        float pms = 0.f;
        if (ctx->fl_packs) PK_HIP(ctx, hipEventElapsedTime(&pms, ctx->ev_p0, ctx->ev_p1));
        stats->pack_ms = pms;
        stats->packs = ctx->fl_packs;
    }
    ctx->fl_packs = 0;
    return 0;
}

int32_t pk_execute_twe_report(pk_ctx* ctx, int64_t* found, int32_t cap, int32_t* n_found, uint8_t* listed_hit, int32_t n_listed) {
    if (!ctx || !n_found || cap < 0 || n_listed < 0 || (cap > 0 && !found) || (n_listed > 0 && !listed_hit)) return -2;
    if (ctx->in_flight) return ctx->fail("pk_execute_twe_report: a launch is in flight (call pk_execute_end)");
    const int32_t nf = (int32_t)ctx->twe_found_host.size();
    *n_found = ctx->twe_overflow_host ? -1 : nf;  // -1: more distinct keys than the set holds (trust pk_exec_stats.first_time_error_key only)
    for (int32_t k = 0; k < nf && k < cap; k++) found[k] = ctx->twe_found_host[k];
    for (int32_t k = 0; k < n_listed; k++) listed_hit[k] = k < (int32_t)ctx->twe_hit_host.size() ? ctx->twe_hit_host[k] : 0;
    return 0;
}

int32_t pk_generic_variant(pk_ctx* ctx, const pk_exec_params* prm, int32_t sample_flags, int32_t nsample, const int32_t* sample_fids, int32_t* key,
                           int32_t* lds, int32_t* typed, int32_t* fast) {
    if (!ctx || !prm || !key || !lds || !typed || !fast || (nsample > 0 && !sample_fids)) return -2;
    KArgs a;
    size_t lds_bytes = 0;
    int use_lds = 0;
    const int32_t rc = fill_args(ctx, prm, a, lds_bytes, use_lds);
    if (rc) return rc;
    const int ik = prm->interp_uv >= 2 ? 2 : prm->interp_uv;
    *key = (ctx->fields[prm->fU].d.dtype == PK_F32 ? 6 : 0) + (ctx->grids[a.main_grid].d.kind == 1 ? 3 : 0) + ik;
    *lds = use_lds;
    *typed = ctx_is_typed(ctx) ? 1 : 0;
    *fast = 0;
    const int32_t sflags = PK_USER_RIDE | (sample_flags & (PK_USER_SAMPLES_UV | PK_USER_SAMPLES_UVW));
    const bool samples = nsample > 0 || (sample_flags & (PK_USER_SAMPLES_UV | PK_USER_SAMPLES_UVW));
    const int ufast = (use_lds && !*typed) ? user_fast_shape(prm, sflags) : -1;
    if (ufast >= 0 && ctx->grids[a.main_grid].d.kind != 1) {
        const int32_t rc2 = fill_fast(ctx, prm, a, ufast == 1);
        if (rc2) return rc2;
        if (a.fast.ok && fill_fast_scalars(ctx, prm, a.fast, nsample, sample_fids)) *fast = 1 + ufast;
    } else if (ufast >= 0 && !samples) {  // the dedicated curvilinear C-grid kernel (a first launch without `ei` guesses still runs the interpreter)
        pk_exec_params guessed = *prm;
        guessed.have_guess0 = 1;
        size_t cl = 0;
        const int32_t rc2 = fill_fastc(ctx, &guessed, a, ufast == 1, cl);
        if (rc2) return rc2;
        if (a.fastc.ok) *fast = 3 + ufast;
    }
    return 0;
}
int32_t pk_set_user_program(pk_ctx* ctx, void* launcher, int32_t flags, int32_t nsample, const int32_t* sample_fids) {
    if (!ctx) return -2;
    if (ctx->in_flight) return ctx->fail("pk_set_user_program: a launch is in flight (call pk_execute_end)");
    if (nsample < 0 || nsample > 4 || (nsample > 0 && !sample_fids)) return ctx->fail("pk_set_user_program: 0 .. 4 sampled scalar fields");
    ctx->user_launch = (void (*)(const void*, int32_t, int32_t, int32_t, uint64_t, void*))launcher;
    ctx->user_flags = launcher ? flags : 0;
    ctx->user_nsample = launcher ? nsample : 0;
    for (int k = 0; k < 4; k++) ctx->user_sample_fid[k] = (launcher && k < nsample) ? sample_fids[k] : 0;
    return 0;
}

int32_t pk_execute_rerun(pk_ctx* ctx, int32_t max_iters, pk_exec_stats* stats) {
    if (!ctx) return -2;
    if (ctx->in_flight) return ctx->fail("pk_execute_rerun: a launch is in flight (call pk_execute_end)");
    if (!ctx->rerun_valid) return ctx->fail("pk_execute_rerun: the state before the last launch is gone (it must directly follow pk_execute / pk_execute_end)");
    if (max_iters < 1) return ctx->fail("pk_execute_rerun: max_iters must be >= 1");
    swap_launch_outputs(ctx);  // `dev` is the state the launch read again (cell-sorted if it sorted: the permutation is unchanged)
    pk_exec_params prm = ctx->rerun_prm;
    prm.sort_by_cell = 0;
    prm.max_iters = max_iters;
    prm.twe_key = ctx->rerun_keys.data();  // (twe_n is the launch's)
    int32_t rc = pk_execute_begin(ctx, &prm);
    if (rc) return rc;
    return pk_execute_end(ctx, stats);
}

int32_t pk_execute_rerun_keys(pk_ctx* ctx, int32_t max_iters, int32_t n_keys, const int64_t* keys, pk_exec_stats* stats) {
    if (!ctx) return -2;
    if (ctx->in_flight) return ctx->fail("pk_execute_rerun_keys: a launch is in flight (call pk_execute_end)");
    if (!ctx->rerun_valid) return ctx->fail("pk_execute_rerun_keys: the state before the last launch is gone (it must directly follow pk_execute / pk_execute_end)");
    if (max_iters < 0) return ctx->fail("pk_execute_rerun_keys: max_iters must be >= 0");
    if (n_keys < 0 || n_keys > PK_MAX_TWE || (n_keys > 0 && !keys)) return ctx->fail("pk_execute_rerun_keys: 0 .. PK_MAX_TWE keys");
    swap_launch_outputs(ctx);  // (see pk_execute_rerun)
    pk_exec_params prm = ctx->rerun_prm;
    prm.sort_by_cell = 0;
    prm.max_iters = max_iters;
    prm.twe_n = n_keys;
    prm.twe_key = keys;
    int32_t rc = pk_execute_begin(ctx, &prm);
    if (rc) return rc;
    return pk_execute_end(ctx, stats);
}

int32_t pk_execute(pk_ctx* ctx, const pk_exec_params* prm, pk_exec_stats* stats) {
    int32_t rc = pk_execute_begin(ctx, prm);
    if (rc) return rc;
    return pk_execute_end(ctx, stats);
}

// ---- sampling ------------------------------------------------------------------------------------------
int32_t pk_eval(pk_ctx* ctx, const pk_exec_params* prm, int32_t what, int64_t m, const double* t, const double* z,
                const double* y, const double* x, double* out_u, double* out_v, double* out_w, int32_t* out_state) {
    if (!ctx || !prm || !t || !z || !y || !x || !out_u) return -2;
    PK_HIP(ctx, hipSetDevice(ctx->device));
    if (m <= 0) return 0;
    pk_exec_params p2 = *prm;
    p2.twe_n = 0;
    p2.reset_state = ctx->eval_points_f32 ? 1 : 0;  // read by eval_kernel as "sample points are float32 columns"
    if (what >= 0) {  // scalar sampling: the main grid is the sampled field's grid
        if (what >= (int)ctx->fields.size()) return ctx->fail("unknown field id");
        if (p2.fU < 0) { p2.fU = what; p2.fV = what; }
    }
    DParticles saved = ctx->dev;
    const bool was_bound = ctx->bound;
    KArgs a;
    size_t lds_bytes;
    int use_lds;
    ctx->bound = true;
    int32_t rc = fill_args(ctx, &p2, a, lds_bytes, use_lds);
    ctx->bound = was_bound;
    ctx->dev = saved;
    if (rc) return rc;
    a.win_lo = -INFINITY;
    a.win_hi = INFINITY;
    double* d = nullptr;
    int32_t* ds = nullptr;
    PK_HIP(ctx, hipMalloc((void**)&d, sizeof(double) * m * 7));
    PK_HIP(ctx, hipMalloc((void**)&ds, sizeof(int32_t) * m));
    double *dt_ = d, *dz = d + m, *dy = d + 2 * m, *dx = d + 3 * m, *du = d + 4 * m, *dv = d + 5 * m, *dw = d + 6 * m;
    PK_HIP(ctx, hipMemcpyAsync(dt_, t, sizeof(double) * m, hipMemcpyHostToDevice, ctx->compute));
    PK_HIP(ctx, hipMemcpyAsync(dz, z, sizeof(double) * m, hipMemcpyHostToDevice, ctx->compute));
    PK_HIP(ctx, hipMemcpyAsync(dy, y, sizeof(double) * m, hipMemcpyHostToDevice, ctx->compute));
    PK_HIP(ctx, hipMemcpyAsync(dx, x, sizeof(double) * m, hipMemcpyHostToDevice, ctx->compute));
    const dim3 grid((unsigned)((m + 255) / 256));
    const int fsel = what >= 0 ? what : p2.fU;
    const bool f32 = ctx->fields[fsel].d.dtype == PK_F32;
    if ((what >= 0 && ctx->fields[fsel].d.is_const == 4) || (what < 0 && p2.interp_uv >= 2)) {  // batch-global lenT / lenZ
        unsigned* dflags = (unsigned*)ds;  // reused as the state output afterwards
        PK_HIP(ctx, hipMemsetAsync(dflags, 0, sizeof(unsigned), ctx->compute));
        hipLaunchKernelGGL(batch_len_kernel, grid, dim3(256), 0, ctx->compute, ctx->fields[fsel].d, ctx->grids[ctx->fields[fsel].d.grid].d, m, dt_, dz, dflags);
        unsigned hflags = 0;
        PK_HIP(ctx, hipMemcpyAsync(&hflags, dflags, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->compute));
        PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
        a.prm.force_lent = (hflags & 1u) ? 2 : 1;
        a.prm.force_lenz = (hflags & 2u) ? 2 : 1;
    }
    const bool typed = ctx_is_typed(ctx);
#define PK_EVAL(FT, IN, TY) hipLaunchKernelGGL((eval_kernel<FT, IN, TY>), grid, dim3(256), 0, ctx->compute, a, what, m, dt_, dz, dy, dx, du, dv, dw, ds)
#define PK_EVAL_T(FT, IN) do { if (typed) PK_EVAL(FT, IN, true); else PK_EVAL(FT, IN, false); } while (0)
    const int ik = p2.interp_uv >= 2 ? 2 : p2.interp_uv;
    if (f32) {
        if (ik == 2) PK_EVAL_T(float, 2); else if (ik == 1) PK_EVAL_T(float, 1); else PK_EVAL_T(float, 0);
    } else {
        if (ik == 2) PK_EVAL_T(double, 2); else if (ik == 1) PK_EVAL_T(double, 1); else PK_EVAL_T(double, 0);
    }
#undef PK_EVAL_T
#undef PK_EVAL
    PK_HIP(ctx, hipGetLastError());
    PK_HIP(ctx, hipMemcpyAsync(out_u, du, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->compute));
    if (out_v) PK_HIP(ctx, hipMemcpyAsync(out_v, dv, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->compute));
    if (out_w) PK_HIP(ctx, hipMemcpyAsync(out_w, dw, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->compute));
    if (out_state) PK_HIP(ctx, hipMemcpyAsync(out_state, ds, sizeof(int32_t) * m, hipMemcpyDeviceToHost, ctx->compute));
    PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
    PK_HIP(ctx, hipFree(d));
    PK_HIP(ctx, hipFree(ds));
    return 0;
}

int32_t pk_search(pk_ctx* ctx, int32_t grid_id, int64_t m, const double* z, const double* y, const double* x, int32_t* ei_out) {
    if (!ctx || !z || !y || !x || !ei_out) return -2;
    if (grid_id < 0 || grid_id >= (int)ctx->grids.size()) return ctx->fail("unknown grid id");
    PK_HIP(ctx, hipSetDevice(ctx->device));
    if (m <= 0) return 0;
    double* d = nullptr;
    int32_t* de = nullptr;
    PK_HIP(ctx, hipMalloc((void**)&d, sizeof(double) * m * 3));
    PK_HIP(ctx, hipMalloc((void**)&de, sizeof(int32_t) * m));
    PK_HIP(ctx, hipMemcpyAsync(d, z, sizeof(double) * m, hipMemcpyHostToDevice, ctx->compute));
    PK_HIP(ctx, hipMemcpyAsync(d + m, y, sizeof(double) * m, hipMemcpyHostToDevice, ctx->compute));
    PK_HIP(ctx, hipMemcpyAsync(d + 2 * m, x, sizeof(double) * m, hipMemcpyHostToDevice, ctx->compute));
    hipLaunchKernelGGL(search_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->compute, ctx->grids[grid_id].d, m, d, d + m, d + 2 * m, de);
    PK_HIP(ctx, hipGetLastError());
    PK_HIP(ctx, hipMemcpyAsync(ei_out, de, sizeof(int32_t) * m, hipMemcpyDeviceToHost, ctx->compute));
    PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
    PK_HIP(ctx, hipFree(d));
    PK_HIP(ctx, hipFree(de));
    return 0;
}

int32_t pk_measure_copy_bandwidth(pk_ctx* ctx, int64_t bytes, int32_t iters, double* gbps) {
    if (!ctx || !gbps || bytes < 16 || iters < 1) return -2;
    PK_HIP(ctx, hipSetDevice(ctx->device));
    void *src = nullptr, *dst = nullptr;
    const int64_t n16 = bytes / 16;
    PK_HIP(ctx, hipMalloc(&src, n16 * 16));
    PK_HIP(ctx, hipMalloc(&dst, n16 * 16));
    PK_HIP(ctx, hipMemsetAsync(src, 1, n16 * 16, ctx->compute));
    const unsigned grid = (unsigned)((n16 + 1023) / 1024);
    hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, ctx->compute, (const float4*)src, (float4*)dst, n16);
    PK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->compute));
    for (int k = 0; k < iters; k++)
        hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, ctx->compute, (const float4*)src, (float4*)dst, n16);
    PK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->compute));
    PK_HIP(ctx, hipStreamSynchronize(ctx->compute));
    float ms = 0.f;
    PK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *gbps = (2.0 * (double)n16 * 16.0 * iters) / ((double)ms * 1e-3) / 1e9;
    PK_HIP(ctx, hipFree(src));
    PK_HIP(ctx, hipFree(dst));
    return 0;
}

}  // extern "C"

#include "pk_comm.inc"
