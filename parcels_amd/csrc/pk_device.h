// pk_device.h -- device-side numerics of the MI355X particle engine (gfx950 only, HIP).
//
// One lane == one particle.  Everything a particle needs for a field evaluation happens in registers:
// time search, 1-D / curvilinear cell search, corner gather, interpolation, status-code update.
// The arithmetic follows the reference expression by expression (file:line cited per function, paths
// relative to /root/reference/src/parcels) and the library is built with -ffp-contract=off because
// NumPy never fuses a*b+c; that is what makes fp64 trajectories agree with the reference to rounding.
//
// This file is written from scratch for CDNA4; it shares no code with oracle/ (the checker).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/parcels_hip.h"

namespace pk {

// ---- device descriptors (built on the host from pk_grid_desc / pk_field_desc) ------------------------
struct DGrid {
    int32_t kind, spherical;
    int32_t has_x, has_y, has_z;
    int32_t nx, ny, nz;
    int32_t xdim, ydim, zdim;
    int32_t off_x, off_y, off_z;
    int32_t lon_f32, lat_f32, depth_f32;
    int32_t h_bitwidth;
    double deg2m;
    const double* lon;
    const double* lat;
    const double* depth;
    const double* node_tab;  // curvilinear: array-of-structs {lon, lat, X, Y, Z} per node (X,Y,Z unit sphere; 0 on a flat mesh):
                             // the 2 nodes of a cell row are 80 contiguous bytes -> 2 rows = 2-4 cache lines instead of 10
    // curvilinear, optional (NULL = compute from node_tab): per cell (yi * nx + xi) CT_STRIDE doubles -- {lon, lat} of the four corners,
    // the tangent-plane basis and projected corners of _spherical_project_cell_and_query (everything that does not depend on the query
    // point), and the cell's quantised hash box.  Built once per grid by the same device functions the search would run per miss:
    // 2.5 GB for a 4322 x 3059 mesh, a trade the 288 GB of HBM make cheap.
    const double* cell_tab;
    const uint32_t* h_keys;
    const int64_t* h_starts;
    const int64_t* h_counts;
    const uint32_t* h_faces;
    int64_t h_nkeys;
    int32_t walk_ok;       // the mesh has no coincident nodes: neighbour-first probing is exact (curvilinear_search)
    const int32_t* h_dir;  // directory over h_keys by the top bits of the code (pk_hashbuild.h), NULL = none
    int32_t h_dir_shift;
    double h_bbox[6];
    double h_rinv[3];  // RN(1 / (max - min)) of the hash grid per axis, 0 where the width is not well scaled (quantize)
    double zfirst, zlast, yfirst, ylast, xfirst, xlast;  // first/last of the 1-D coordinate vectors (rectilinear; depth always)
};

struct DField {
    int32_t grid, dtype;
    int32_t nt, nz, ny, nx;
    int32_t has_time_interval, is_const;  // is_const: scalar interpolator 0 XLinear 1 XConstantField 2 XNearest 3 CGrid_Tracer 4 InvdistLandTracer
    int32_t nslots, pad;
    int32_t ncomp, comp;             // component packing: element (slot, z, y, x) of this field lives at (off * ncomp + comp)
    int64_t st_t, st_z, st_y, st_x;  // element strides (0 for axes the field does not have)
    const void* data;                // nslots * st_t elements (ring of time levels)
    const double* time;              // nt
    double tlen;                     // time[nt-1] - time[0]
    double tfirst, tlast;            // time[0], time[nt-1]
};

struct DParticles {
    int64_t n;
    int32_t ngrids, spatial_f32;
    double* t;
    void *z, *y, *x, *dz, *dy, *dx;
    double* dt;
    double* next_dt;
    int32_t* state;
    int32_t* ei;
    int64_t* particle_id;
    void* extra[PK_MAX_EXTRA];        // user Variables written by device kernels (PK_KERNEL_SAMPLE_FIELD)
    int32_t extra_f32[PK_MAX_EXTRA];  // 1: float32 column, 0: float64
    int32_t* iter;                    // device only: loop iterations (kernel.py:190) the particle has made since this Kernel.execute call began
};

// Where a launch WRITES the particle state.  The advection kernels read a particle from DParticles (KArgs::p) and write it here:
// the host points this at the second column set, so the set a launch read stays intact -- the state before the launch, for free --
// and the launch can be repeated with an iteration limit when some particle raised an error (pk_execute_rerun, kernel.py:236-245:
// the reference stops every particle after the iteration in which the first one erred).  particle_id and the user Variables are
// not part of it (never written / rewritten by every iteration).
struct DPOut {
    double* t;
    void *z, *y, *x, *dz, *dy, *dx;
    double* dt;
    double* next_dt;
    int32_t* state;
    int32_t* ei;
    int32_t* iter;
};

struct DCounters {
    unsigned long long steps, attempts, paused;
    unsigned int err_iter, pad;  // smallest iteration index (1-based) in which a particle entered an error state; 0xFFFFFFFF = none
    unsigned long long twe_key;  // smallest key of a sample (pk_exec_params.twe_key) at which a particle left a time interval; ~0 = none
    unsigned int twe_overflow, pad2;  // keys that found no slot in KArgs::twe_found (the host then trusts only twe_key)
};

// Wave-uniform constants of the fast path for XLinear_Velocity on a rectilinear A-grid with float64 coordinates
// (pk_fast_agrid.h), folded from the grid / field descriptors by the host (pk_api.hip: fill_fast).
struct FastA {
    int32_t ok, grid;                           // preconditions hold; grid id (column of `ei`)
    int32_t has_ti, has_z, has_y, has_x, spherical;
    int32_t nt, nslots;                         // time levels of U / V / W and their ring
    int32_t gnz, gny, gnx;                      // node counts of the grid axes
    uint32_t ex, ey, ez;                        // ravel strides of `ei` (basegrid.py:83-152), 0 for an axis the grid lacks
    uint32_t st_z, st_y;                        // element strides of the fields inside a level (st_x == 1)
    uint32_t dyb, dzb;                          // byte strides to the yi+1 row / zi+1 plane, 0 if the fields have no such neighbour
    int32_t lds_time, lds_depth, lds_lat, lds_lon, lds_n;  // offsets (in pairs) of the {a, 1/width} tables inside `tab`
    int32_t lds_blk;                            // offset (in pairs) of the per-lane corner-block cache behind the tables, 0 = none (pk_fast_agrid.h)
    int64_t lvl_b;                              // bytes per time level (< 2^32)
    const char *U, *V, *W;                      // level rings (W may be NULL)
    const double* tab;                          // global copy of the interleaved coordinate tables: time | depth | lat | lon
    double tlen, t0, t1, z0, z1, y0, y1, x0, x1;
    double deg2m, inv_deg2m;
    // scalar fields a compiled user kernel samples while it rides in the dedicated kernel (parcels_amd/jit.py): same grid, dtype, layout,
    // time axis and ring as U -- only the base pointer differs -- and XLinear; sfid: their field ids (what Request::fidx names)
    int32_t ns, pad1;
    int32_t sfid[4];
    const char* S[4];
};

// Wave-uniform constants of the fast path for CGrid_Velocity on a spherical curvilinear C-grid with float64 node coordinates
// (pk_fast_cgrid.h), folded from the grid / field descriptors by the host (pk_api.hip: fill_fastc).
struct FastC {
    int32_t ok, grid;                    // preconditions hold; grid id (column of `ei`)
    int32_t has_ti, has_z, walk_ok, near_edges;  // near_edges: no cell spans 2^-8 rad of latitude (pk_fast_cgrid.h: cos_near)
    int32_t nt, nslots;                  // time levels of U / V / W and their ring
    int32_t gnz, gny, gnx;               // node counts of the grid axes
    uint32_t ex, ey, ez;                 // ravel strides of `ei` (basegrid.py:83-152)
    int32_t lds_time, lds_depth, lds_n;  // offsets (in pairs) of the {a, 1/width} tables of time | depth inside `tab`
    int32_t lds_rec, lds_fv;             // offsets (in doubles) of the per-lane cell slots inside the dynamic LDS
    int32_t st_z, st_y;                  // element strides of the fields inside a level (cells)
    int32_t cb;                          // bytes per cell struct of the (possibly packed) field buffers
    int64_t lvl_b;                       // bytes per time-level slot
    // byte offsets of the six staggered values relative to the struct of cell (zi, yi, xi); level rings (component offset folded into
    // dU0.. for packed groups).  The 2-D kernels have no W: its three members carry, for them, the CELL-PACKED PAIR COPY of the
    // staggered values -- one group per cell and pair of adjacent levels, {U0, U1, V0, V1}(level L), {U0, U1, V0, V1}(level L + 1), so
    // that a cell change reads ONE line for both levels instead of four (pk_api.hip: ensure_velocity_pairs; pair L lives in slot
    // L % nslots; vp == NULL: not available, read the level rings).  Same members, so the kernel arguments keep their size and the 3-D
    // and A-grid kernels their code.
    int64_t dU0, dU1, dV0, dV1;
    union { int64_t dW0; int64_t vp_slot_b; };  // 2-D: bytes per pair slot
    union { int64_t dW1; int64_t vp_hi; };      // 2-D: highest resident level -- a sample exactly ON it reads the upper half of the pair below
    const char *U, *V;
    union { const char* W; const char* vp; };   // W may be NULL
    // AdvectionDiffusionM1: the two scalar fields Kh_zonal / Kh_meridional on the nodes of the SAME grid (XLinear): base, byte strides of
    // their axes (0 for an axis the field does not have), extents, and whether they share the velocity's time axis (else: no time axis)
    const char* kh[2];
    int32_t kh_st[2], kh_sz[2], kh_sy[2];   // bytes per time level / depth level / row
    int32_t kh_nt[2], kh_nz[2], kh_ny[2], kh_nx[2], kh_has_ti[2], kh_nslots[2];
    const double* ct2;                   // per-cell records (pk_fast_cgrid.h: CT2_STRIDE doubles each)
    const double* tab;                   // global copy of the interleaved coordinate tables: time | depth
    double tlen, t0, t1, z0, z1, deg2m;
};

// The grid / field descriptors of a context live in ONE device buffer (uploaded before a launch when they changed), not in the kernel
// arguments: the kernarg segment is 4 KiB, and 4 grids + 16 fields used 3 KiB of it -- one more descriptor member and the design had to
// change (and any code shape that made the compiler take the address of the by-value argument cost a 4 KiB private copy per lane).
// The pointers are in the CONSTANT address space: uniform loads through them are scalar loads, hoistable like kernarg loads.
#define PK_CONST_AS __attribute__((address_space(4)))
struct KArgs {
    const PK_CONST_AS DGrid* grids;    // [number of grids of the context]   (kgrid / kfield below)
    const PK_CONST_AS DField* fields;  // [number of fields of the context]
    DParticles p;
    pk_exec_params prm;
    double win_lo, win_hi;  // resident time window of the time-varying fields (seconds)
    DCounters* counters;
    const unsigned long long* twe_listed;  // the launch's copy of pk_exec_params.twe_key (ascending; device memory, read on the cold path only)
    unsigned long long* twe_found;         // hash set (TWE_FOUND_SLOTS slots, 0 = empty) of the unlisted samples at which a lane left a time interval
    unsigned int* twe_hit;                 // [twe_n] listed sample k: some lane really was outside the interval there (the listing is justified)
    // LDS layout of the main grid's 1-D arrays (element offsets into the dynamic shared array, -1 = global)
    int32_t lds_time, lds_depth, lds_lat, lds_lon, lds_total;
    int32_t lds_cc_nodes, lds_cc_keys, lds_cc_fvals;  // cell cache (CellCache) offsets in doubles from the LDS base, -1 = off
    int32_t main_grid, main_field;
    DPOut po;  // output columns of the launch (== the columns of `p` for an in-place launch)
    union {  // at most one of the dedicated kernels runs per launch
        FastA fast;
        FastC fastc;
    };
};

// ---- small helpers ------------------------------------------------------------------------------------
#define PK_DEV __device__ __forceinline__
// descriptor g / f of the launch (the cast to the generic address space is undone by address-space inference once inlined)
PK_DEV const DGrid& kgrid(const KArgs& a, int g) { return *(const DGrid*)(a.grids + g); }
PK_DEV const DField& kfield(const KArgs& a, int f) { return *(const DField*)(a.fields + f); }
// The call-wide OutsideTimeInterval (index_search.py:85-86 raises for the whole call; field.py:31-44 writes the code into every particle of
// the view): a sample of a Kernel.execute call is named by (iteration `it`, kernel slot * 1000 + sample number `klo`), key = it << 32 | klo.  twe_listed: the
// host knows that SOME particle leaves the field's time interval at this sample -- every particle that reaches it takes code 70 and the
// value 0, nothing else of the call is written.  twe_note: this lane just left a time interval at a sample that is not listed; the
// smallest such key of the launch goes back to the host, which repeats the call with it (include/parcels_hip.h: pk_exec_params.twe_key).
// it == 0: not inside the loop of kernel.py:190 (pk_eval, body_only launches: the caller owns the batch there).
PK_DEV unsigned long long twe_sample_key(unsigned it, int klo) { return ((unsigned long long)it << 32) | (unsigned long long)(unsigned)klo; }
PK_DEV bool twe_listed(const KArgs& a, unsigned it, int klo) {
    const int n = a.prm.twe_n;  // wave-uniform: one scalar compare per sample when nothing is listed
    if (__builtin_expect(n == 0, 1)) return false;
    const unsigned long long key = twe_sample_key(it, klo);
    const unsigned long long* lst = a.twe_listed;  // ascending: bisection
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (lst[mid] < key) lo = mid + 1; else hi = mid;
    }
    return it != 0 && lo < n && lst[lo] == key;
}
PK_DEV void twe_note(const KArgs& a, unsigned it, int klo) {
    if (it != 0) atomicMin(&a.counters->twe_key, twe_sample_key(it, klo));
}
// The general programs report EVERY unlisted sample at which a lane left a time interval, not only the smallest (twe_note, which is all the
// dedicated kernels do: their hot loops carry nothing else): a particle past the last time level fails every later sample as well, and its
// own trajectory does not depend on whether those samples are listed -- it takes code 70 and zeros either way -- so the host can list them all
// at once and VALIDATE the listing with the next pass (twe_justify) instead of finding one key per pass (DeviceEngine.execute).
constexpr int TWE_FOUND_SLOTS = 4096;  // open addressing, linear probing; counters->twe_overflow counts the keys that found no slot
PK_DEV void twe_note_all(const KArgs& a, unsigned it, int klo) {
    if (it == 0) return;
    const unsigned long long key = twe_sample_key(it, klo);
    atomicMin(&a.counters->twe_key, key);
    unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 52);  // 12 bits
    for (int probe = 0; probe < TWE_FOUND_SLOTS; probe++) {
        unsigned long long* slot = a.twe_found + ((h + probe) & (TWE_FOUND_SLOTS - 1));
        unsigned long long cur = *(volatile unsigned long long*)slot;
        if (cur == key) return;
        if (cur == 0ull) {
            cur = atomicCAS(slot, 0ull, key);
            if (cur == 0ull || cur == key) return;
        }
    }
    atomicAdd(&a.counters->twe_overflow, 1u);
}
// index of the sample in the launch's list, -1 = not listed
PK_DEV int twe_listed_index(const KArgs& a, unsigned it, int klo) {
    const int n = a.prm.twe_n;  // wave-uniform: one scalar compare per sample when nothing is listed
    if (__builtin_expect(n == 0, 1)) return -1;
    const unsigned long long key = twe_sample_key(it, klo);
    const unsigned long long* lst = a.twe_listed;  // ascending: bisection
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (lst[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (it != 0 && lo < n && lst[lo] == key) ? lo : -1;
}
// a lane at listed sample `idx` that IS outside the interval [0, tlen]: the listing stands (plain store of a constant: benign race)
PK_DEV void twe_justify(const KArgs& a, int idx, double t, double tlen) {
    if (!(0 <= t) || !(t <= tlen)) a.twe_hit[idx] = 1u;
}

// Scheduling fence between the gathers of two fields: without it the compiler hoists all 16 (32, 48) corner loads of
// U, V (and W) above the first interpolation, which costs ~64 VGPRs per field and caps occupancy at 2 waves/SIMD.
#ifndef PK_FIELD_FENCE
#define PK_FIELD_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

PK_DEV int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
PK_DEV int mini(int a, int b) { return a < b ? a : b; }

// NumPy's sort order puts NaN last; searchsorted uses it (index_search.py:47)
PK_DEV bool np_less(double a, double b) { return a < b || (b != b && a == a); }

static constexpr double DEG2RAD = 3.14159265358979323846 / 180.0;          // npy_deg2rad: x*(NPY_PI/180.0)
static constexpr float DEG2RADF = 3.14159265358979323846f / 180.0f;        // float32 ufunc loop
static constexpr int GRID_SEARCH_ERROR = -3;                                // index_search.py:15-17
static constexpr int LEFT_OUT_OF_BOUNDS = -2;
static constexpr int RIGHT_OUT_OF_BOUNDS = -1;

// cos() for latitudes.  |x| <= 1.5 rad (86 deg): Cody-Waite reduction by pi/2 + the classic minimax kernels
// (error < 0.82 ulp, checked against long-double cosl over 2e7 samples); beyond that sincos_geo.  The reference's
// cos is NumPy/libm (<= 1 ulp): any <= 1 ulp cosine is as close to it as another libm would be.  ~20 fp64 ops
// instead of the generic routine's range reduction.
PK_DEV void sincos_geo(double x, double& s, double& c);
PK_DEV double cos_lat(double x) {
    const double ax = fabs(x);
    if (ax > 1.5) {  // beyond +-86 degrees (or a stray particle): general reduction, still no library call
        double s_, c_;
        sincos_geo(x, s_, c_);
        return c_;
    }
    if (ax <= 0.78539816339744830962) {
        const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                     C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
        const double z = ax * ax;
        const double r = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
        if (ax < 0.3) return 1.0 - (0.5 * z - z * r);
        double qx = ax > 0.78125 ? 0.28125 : ax * 0.25;
        qx = __longlong_as_double(__double_as_longlong(qx) & 0xffffffff00000000ll);
        const double hz = 0.5 * z - qx, a_ = 1.0 - qx;
        return a_ - (hz - z * r);
    }
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double zz = ax - pio2_1;
    const double y0 = zz - pio2_1t;
    const double y1 = (zz - y0) - pio2_1t;
    const double z = y0 * y0, v = z * y0;
    const double r = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
    return -(y0 - ((z * (0.5 * y1 - v * r) - y1) - v * S1));
}

struct GPos {
    int ti, zi, yi, xi;
    double tau, zeta, eta, xsi;
    bool w32;  // xsi, eta are float32 ARRAYS in the reference: a curvilinear evaluation without any guess takes them straight from
               // the hash query's float32 buffer (spatialhash.py:505), so NumPy forms every expression made of xsi, eta and
               // Python scalars alone in float32 (1 - xsi, (1 - xsi) * (1 - eta), ...) before it meets float64 data
    bool x32, e32, z32;  // the bcoord ARRAY of that axis is float32 in the reference: w32, or a float32 coordinate searched with
                         // float32 particle positions (index_search.py:51: f32 - f32 stays f32)
    double xsi_raw, eta_raw;  // curvilinear: the float64 (xsi, eta) of the point in cell (yi, xi) BEFORE a hash hit is rounded to float32
                              // (spatialhash.py:505), i.e. what a search started in that cell returns (index_search.py:269-285)
};

// ---- exact divisions that cost less than v_div_scale / v_rcp / v_div_fmas / v_div_fixup per quotient ----------------------
// (1) n / d with r = RN(1/d) computed in IEEE arithmetic on the host: q0 = RN(n r) is within a few ulp, the fused residual
// e0 = n - d q0 is exact up to a relative 2^-53 of itself, q1 = RN(q0 + e0 r) is faithful; a second residual step then yields the
// correctly rounded quotient (Markstein 1990; Thm. 8.5 of Muller et al., Handbook of Floating-Point Arithmetic: y = RN(1/b),
// q faithful, r = RN(a - bq) exact => RN(q + r y) = RN(a/b)).  Quotients that come out zero, tiny or NaN take the hardware
// division (the exactness argument needs normal numbers; the denominators are checked by the host to lie in [1e-100, 1e100], so
// a huge quotient means a huge numerator, which the residuals handle).
PK_DEV double div_by_recip(double n, double d, double r) {
    const double q0 = n * r;
    const double e0 = __builtin_fma(-d, q0, n);
    const double q1 = __builtin_fma(e0, r, q0);
    const double e1 = __builtin_fma(-d, q1, n);
    double q = __builtin_fma(e1, r, q1);
    // zero, subnormal-range and NaN quotients (an infinite numerator or an overflowing product ends as NaN above)
    if (__builtin_expect(!(fabs(q) >= 1e-250), 0)) {
        // rare (a sample point exactly on a node, t exactly on a time level, non-finite input).  The empty volatile asm keeps the
        // optimiser from if-converting this block: speculated, the 14-instruction hardware division would run on every call
        double nn = n;
        asm volatile("" : "+v"(nn));
        q = nn / d;
    }
    return q;
}
// (2) several quotients over ONE run-time denominator: the compiler expands every fp64 `/` into
//   div_scale x2, rcp, fma x4 (two Newton steps on the reciprocal), mul, fma, div_fmas, div_fixup
// (AMDGPU LowerFDIV64).  The reciprocal part depends on the denominator only: formed once here with the same instructions, then
// each quotient is the remaining mul + fma + fma -- the identical operation sequence, hence the identical (IEEE) result, as long
// as div_scale would not have rescaled anything: |d| in [1e-100, 1e100] and the quotient in [1e-150, 1e150] (then |n| is in
// [1e-250, 1e250], far from every rescaling rule of v_div_scale_f64); anything else takes the hardware division.
struct Recip {
    double d, r;
    bool ok;
};
PK_DEV Recip make_recip(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double ad = fabs(d);
    return Recip{d, r, ad >= 1e-100 && ad <= 1e100};
}
PK_DEV double div_shared(double n, const Recip& R) {
    const double q = n * R.r;
    const double e = __builtin_fma(-R.d, q, n);
    double res = __builtin_fma(e, R.r, q);
    const double ar = fabs(res);
    if (__builtin_expect(!(R.ok && ar >= 1e-150 && ar <= 1e150), 0)) {
        double nn = n;
        asm volatile("" : "+v"(nn));  // keep the rare path a branch (see div_by_recip)
        res = nn / R.d;
    }
    return res;
}

// NumPy dtype propagation for the expressions in which float32 arrays meet: float32 op float32 -> float32 (rounded),
// anything with a float64 array -> float64; Python scalars adapt to the array.  (value, is-float32-array) pairs.
struct TV {
    double v;
    bool f32;
};
PK_DEV TV tv_mul(TV a, TV b) { const bool f = a.f32 && b.f32; return TV{f ? (double)((float)a.v * (float)b.v) : a.v * b.v, f}; }
PK_DEV TV tv_add(TV a, TV b) { const bool f = a.f32 && b.f32; return TV{f ? (double)((float)a.v + (float)b.v) : a.v + b.v, f}; }
PK_DEV TV tv_one_minus(TV a) { return TV{a.f32 ? (double)(1.0f - (float)a.v) : 1 - a.v, a.f32}; }

// The same for whole interpolators (CGrid_Velocity, the slip factors): a value plus the dtype tag of the NumPy array it is in the
// reference -- 0 float64 array, 1 float32 array, 2 Python scalar (adapts to the other operand).  float32 op float32 is a float32
// operation (rounded), anything with a float64 array is float64.
struct NV {
    double v;
    int dt;
};
PK_DEV NV nvv(double v, int dt) { return NV{v, dt}; }
PK_DEV int nv_res(int a, int b) { return a == 2 ? b : (b == 2 ? a : ((a == 1 && b == 1) ? 1 : 0)); }
PK_DEV NV nv_add(NV a, NV b) { const int d = nv_res(a.dt, b.dt); return NV{d == 1 ? (double)((float)a.v + (float)b.v) : a.v + b.v, d}; }
PK_DEV NV nv_sub(NV a, NV b) { const int d = nv_res(a.dt, b.dt); return NV{d == 1 ? (double)((float)a.v - (float)b.v) : a.v - b.v, d}; }
PK_DEV NV nv_mul(NV a, NV b) { const int d = nv_res(a.dt, b.dt); return NV{d == 1 ? (double)((float)a.v * (float)b.v) : a.v * b.v, d}; }
PK_DEV NV nv_div(NV a, NV b) { const int d = nv_res(a.dt, b.dt); return NV{d == 1 ? (double)((float)a.v / (float)b.v) : a.v / b.v, d}; }
PK_DEV NV nv_neg(NV a) { return NV{-a.v, a.dt}; }
PK_DEV NV nv_cast(NV a, int dt) { return NV{dt == 1 ? (double)(float)a.v : a.v, dt}; }
PK_DEV NV nv_sqrt(NV a) { return a.dt == 1 ? NV{(double)sqrtf((float)a.v), 1} : NV{sqrt(a.v), a.dt}; }

// Per-lane LDS cache of the curvilinear cell a particle sits in: its 4 corner nodes {lon, lat, X, Y, Z} and the raw
// staggered field values of one C-grid evaluation at both time levels.  A particle moves a fraction of a cell per RK
// stage, so stages 2..4 and most following steps of the fused loop find their cell here instead of re-fetching ~9
// cache lines per evaluation through L2 (the 1e7 particles of BASELINE config 3 touch ~150 MB per stage, far more
// than the 4 MB L2 of an XCD keeps between stages).  Layout: structure-of-arrays over the 256 lanes of the workgroup
// (element k of lane l at [k * 256 + l]) -> conflict-free ds_read/ds_write.  Cached bits are the global ones, so
// results are unchanged.  All pointers NULL = disabled.
// Workgroup size of the programs that carry the cell cache (curvilinear grids, LDS-staged): ONE wavefront.  The cache costs
// 224 B of LDS per lane (f32 fields); 256-lane workgroups (57 KB) fit twice into the 160 KB of a CU = 2 waves / SIMD whatever the
// register budget, 64-lane workgroups fit ten times, so the occupancy is decided by the registers again.
#ifndef PK_WG_CURV
#define PK_WG_CURV 64
#endif
constexpr int CC_LANES = PK_WG_CURV;
constexpr int CC_NODE_ROWS = 22;
constexpr int CT_STRIDE = 24;  // DGrid::cell_tab record: rows 0..21 as in CellCache::nodes, 22 = the packed quantised box, 23 unused
__host__ __device__ constexpr inline int wg_size(int kind, bool lds) { return (kind == 1 && lds) ? PK_WG_CURV : 256; }
struct CellCache {
    // [CC_NODE_ROWS][CC_LANES]: rows 0..7 {lon, lat} of corner c (0 (yi,xi), 1 (yi,xi+1), 2 (yi+1,xi+1), 3 (yi+1,xi)) at 2c, 2c+1;
    // spherical meshes: rows 8..10 eu, 11..13 ev (the cell's tangent-plane basis), 14..17 / 18..21 the corners projected on it
    // (spherical_project_cell: everything of _spherical_project_cell_and_query that does not depend on the query point)
    double* nodes;
    int* key;       // [4][256]: node cell yi*nx+xi | field cell yi*nx+xi | zi | 4*ti + 2*(W cached) + (level ti+1 cached);  -1 = empty
    void* fvals;    // [12][256] of the field dtype: Ua,Ub,Va,Vb,Wa,Wb at level ti, then at level ti+1 (NULL: not cached)
};

// Coordinate vectors of the main grid, either LDS-staged or global (address space is inferred after inlining).
struct Coords {
    CellCache cc;
    const double* time;
    const double* depth;
    const double* lat;
    const double* lon;
    // first/last value of each vector, read once through the scalar unit (uniform) instead of per lane from LDS
    double t0, t1, z0, z1, y0, y1, x0, x1;
};

// clip(searchsorted(arr, x, "left") - 1, 0, n-2) (index_search.py:47), found by walking from `hint`
// (the cell of the previous evaluation: particles move < 1 cell per stage) with a binary-search fallback.
PK_DEV int cell_index(const double* arr, int n, double x, int hint) {
    if (x != x) return n - 2;  // NaN sorts last
    int i = clampi(hint, 0, n - 2);
    // invariant wanted: (i == 0 || arr[i] < x) && (i == n-2 || !(arr[i+1] < x))
    if (arr[i] < x) {
        int k = 0;
        while (i < n - 2 && arr[i + 1] < x) {
            ++i;
            if (++k == 3) {  // far from the hint: bisect the rest
                int lo = i + 1, hi = n;  // first index in (i, n) with !(arr < x)
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if (arr[mid] < x) lo = mid + 1; else hi = mid;
                }
                i = clampi(lo - 1, 0, n - 2);
                break;
            }
        }
    } else {
        int k = 0;
        while (i > 0 && !(arr[i] < x)) {
            --i;
            if (++k == 3) {
                int lo = 0, hi = i + 1;
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if (arr[mid] < x) lo = mid + 1; else hi = mid;
                }
                i = clampi(lo - 1, 0, n - 2);
                break;
            }
        }
    }
    return i;
}

// _search_1d_array (index_search.py:20-62)
PK_DEV void search_1d(const double* arr, int n, double first, double last, double x, bool arr_f32, bool x_f32, int hint, int& idx,
                      double& bc) {
    if (n < 2) {  // :45-46, no out-of-bounds codes in this branch
        idx = 0;
        bc = 0.0;
        return;
    }
    int i = cell_index(arr, n, x, hint);
    double a0 = arr[i], a1 = arr[i + 1];
    if (arr_f32) {
        float d = (float)a1 - (float)a0;  // f32 array: the width is an f32 subtraction in NumPy
        bc = x_f32 ? (double)(((float)x - (float)a0) / d) : (x - a0) / (double)d;
    } else {
        bc = (x - a0) / (a1 - a0);
    }
    if (x < first) i = LEFT_OUT_OF_BOUNDS;   // :59
    if (x > last) i = RIGHT_OUT_OF_BOUNDS;   // :60
    idx = i;
}

// The two x-corners of a cell are adjacent in memory (x is the fastest axis): fetch them with ONE 16-byte (fp64) /
// 8-byte (fp32) load.  Element alignment is enough: gfx950 global loads handle unaligned dwordx4.
typedef double pk_double2 __attribute__((ext_vector_type(2), aligned(8)));
typedef float pk_float2 __attribute__((ext_vector_type(2), aligned(4)));
typedef float pk_float4 __attribute__((ext_vector_type(4), aligned(16)));
PK_DEV void ldpair(const double* p, double& a, double& b) {
    const pk_double2 v = *reinterpret_cast<const pk_double2*>(p);
    a = v.x;
    b = v.y;
}
PK_DEV void ldpair(const float* p, double& a, double& b) {
    const pk_float2 v = *reinterpret_cast<const pk_float2*>(p);
    a = (double)v.x;
    b = (double)v.y;
}

// ---- curvilinear point-in-cell (index_search.py:94-239, 439-450) -------------------------------------
// np.dot(_invA, p): the integer matrix is promoted to float and every product is formed, also 0*p and 1*p
// (exact for finite p), accumulated left to right.
PK_DEV void bilinear_inverse(const double px[4], const double py[4], double xq, double yq, double& xsi, double& eta) {
    double a0 = px[0];
    double a1 = -px[0] + px[1];
    double a2 = -px[0] + px[3];
    double a3 = ((px[0] + -px[1]) + px[2]) + -px[3];
    double b0 = py[0];
    double b1 = -py[0] + py[1];
    double b2 = -py[0] + py[3];
    double b3 = ((py[0] + -py[1]) + py[2]) + -py[3];
    double aa = a3 * b2 - a2 * b3;
    double bb = a3 * b0 - a0 * b3 + a1 * b2 - a2 * b1 + xq * b3 - yq * a3;
    double cc = a1 * b0 - a0 * b1 + xq * b1 - yq * a1;
    double det2 = bb * bb - 4 * aa * cc;
    double det = det2 > 0 ? sqrt(det2) : -1.0;
    double e;
    if (fabs(aa) < 1e-12) e = -cc / bb;
    else e = det2 > 0 ? (-bb + det) / (2 * aa) : -1.0;
    double x;
    if (fabs(a1 + a3 * e) < 1e-12) x = ((yq - py[0]) / (py[1] - py[0]) + (yq - py[3]) / (py[2] - py[3])) * 0.5;
    else x = (xq - a0 - a2 * e) / (a1 + a3 * e);
    xsi = x;
    eta = e;
}

// sin and cos for angles of geographic size (|x| up to ~1e6 rad): Cody-Waite reduction by n*pi/2 with an 86-bit pi/2
// (n*pio2_1 is exact for |n| < 2^20) + the classic minimax kernels; < 1 ulp like cos_lat.  The library sincos drags
// a Payne-Hanek large-argument path (v_trig_preop) into the kernel that longitudes and latitudes never need.
PK_DEV void sincos_geo(double x, double& s, double& c) {
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double fn = rint(x * 6.36619772367581382433e-01);
    double z0 = x - fn * pio2_1;
    double w = fn * pio2_1t;
    double r = z0 - w;
    if (fabs(fn) > 4.0) {  // second Cody-Waite step (119-bit pi/2) once n*pio2_1t is no longer negligible
        const double pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
        const double t2 = z0;
        w = fn * pio2_2;
        z0 = t2 - w;
        w = fn * pio2_2t - ((t2 - z0) - w);
        r = z0 - w;
    }
    const double t = (z0 - r) - w;  // tail
    const int n = (int)fn;
    const double z = r * r;
    // __kernel_sin(r, t)
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double v = z * r;
    const double rs = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
    const double ks = r - ((z * (0.5 * t - v * rs) - t) - v * S1);
    // __kernel_cos(r, t)
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double rc = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    const double ar = fabs(r);
    double qx = ar > 0.78125 ? 0.28125 : ar * 0.25;
    qx = __longlong_as_double(__double_as_longlong(qx) & 0xffffffff00000000ll);
    if (ar < 0.3) qx = 0.0;
    const double hz = 0.5 * z - qx, a_ = 1.0 - qx;
    const double kc = a_ - (hz - (z * rc - r * t));
    switch (n & 3) {
        case 0: s = ks; c = kc; break;
        case 1: s = kc; c = -ks; break;
        case 2: s = -ks; c = -kc; break;
        default: s = -kc; c = ks; break;
    }
}

PK_DEV void latlon_rad_to_xyz(double lat, double lon, double& X, double& Y, double& Z) {
    double sl, cl, so, co;
    sincos_geo(lat, sl, cl);
    sincos_geo(lon, so, co);
    X = co * cl;
    Y = so * cl;
    Z = sl;
}

// query point on the unit sphere (spherical) or in the plane (flat), computed once per evaluation
struct QPoint {
    double y, x;        // degrees / metres as given
    double qX, qY, qZ;  // unit-sphere coordinates (spherical mesh); (x, y, 0) on a flat mesh
};
PK_DEV QPoint make_qpoint(const DGrid& g, double y, double x) {
    QPoint q;
    q.y = y;
    q.x = x;
    if (g.spherical) latlon_rad_to_xyz(y * DEG2RAD, x * DEG2RAD, q.qX, q.qY, q.qZ);
    else { q.qX = x; q.qY = y; q.qZ = 0.0; }
    return q;
}

// _spherical_project_cell_and_query (index_search.py:180-239).  cX,cY,cZ: unit-sphere coordinates of the 4 corners
// (index_search.py:197-198), read from the per-node table that the host computed once with the reference's own
// expression (cos(lon)cos(lat), sin(lon)cos(lat), sin(lat)) instead of 8 sin/cos pairs per evaluation.
PK_DEV void spherical_project_cell(const double cX[4], const double cY[4], const double cZ[4], double eu[3], double ev[3], double pu[4],
                                   double pv[4]) {
    double ux = (cX[1] + cX[2]) - (cX[0] + cX[3]);
    double uy = (cY[1] + cY[2]) - (cY[0] + cY[3]);
    double uz = (cZ[1] + cZ[2]) - (cZ[0] + cZ[3]);
    double un = sqrt(ux * ux + uy * uy + uz * uz);
    if (un == 0.0) un = 1.0;
    const Recip run = make_recip(un);
    const double eux = div_shared(ux, run), euy = div_shared(uy, run), euz = div_shared(uz, run);
    double vx = (cX[2] + cX[3]) - (cX[0] + cX[1]);
    double vy = (cY[2] + cY[3]) - (cY[0] + cY[1]);
    double vz = (cZ[2] + cZ[3]) - (cZ[0] + cZ[1]);
    double vd = vx * eux + vy * euy + vz * euz;
    vx = vx - vd * eux;
    vy = vy - vd * euy;
    vz = vz - vd * euz;
    double vn = sqrt(vx * vx + vy * vy + vz * vz);
    if (vn == 0.0) vn = 1.0;
    const Recip rvn = make_recip(vn);
    const double evx = div_shared(vx, rvn), evy = div_shared(vy, rvn), evz = div_shared(vz, rvn);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        pu[k] = cX[k] * eux + cY[k] * euy + cZ[k] * euz;
        pv[k] = cX[k] * evx + cY[k] * evy + cZ[k] * evz;
    }
    eu[0] = eux; eu[1] = euy; eu[2] = euz;
    ev[0] = evx; ev[1] = evy; ev[2] = evz;
}
// the query point on the cell's tangent plane
PK_DEV void spherical_project_query(const double eu[3], const double ev[3], const QPoint& q, double& xq, double& yq) {
    xq = q.qX * eu[0] + q.qY * eu[1] + q.qZ * eu[2];
    yq = q.qX * ev[0] + q.qY * ev[1] + q.qZ * ev[2];
}

// ---- Morton spatial hash query (spatialhash.py:389-535, 554-597, 647-765) ------------------------------
PK_DEV uint32_t dilate_bits(uint32_t n) {
    n &= 0x000003FFu;
    n = (n | (n << 16)) & 0xFF0000FFu;
    n = (n | (n << 8)) & 0x0300F00Fu;
    n = (n | (n << 4)) & 0x030C30C3u;
    n = (n | (n << 2)) & 0x09249249u;
    return n;
}
// rinv: RN(1 / (vmax - vmin)) from the host when that width is well scaled (DGrid::h_rinv), else 0 -> hardware division
PK_DEV uint32_t quantize(double v, double vmin, double vmax, int bitwidth, double rinv = 0.0) {
    double d = vmax - vmin;
    double vn = (d != 0) ? (rinv != 0.0 ? div_by_recip(v - vmin, d, rinv) : (v - vmin) / d) : 0.0;
    double q = vn * bitwidth;
    if (!(q >= 0)) q = 0;  // also NaN (rejected by the finite mask anyway)
    if (q > bitwidth) q = bitwidth;
    return (uint32_t)q;
}
// quantised [min, max] of the four corner values of a face along one axis vs the quantised query coordinate
PK_DEV bool in_quantised_box(const DGrid& g, const double c[4], double v, int axis2) {
    const double lo = fmin(fmin(c[0], c[1]), fmin(c[2], c[3])), hi = fmax(fmax(c[0], c[1]), fmax(c[2], c[3]));
    const double ri = g.h_rinv[axis2 >> 1];
    const uint32_t qv = quantize(v, g.h_bbox[axis2], g.h_bbox[axis2 + 1], g.h_bitwidth, ri);
    return quantize(lo, g.h_bbox[axis2], g.h_bbox[axis2 + 1], g.h_bitwidth, ri) <= qv && qv <= quantize(hi, g.h_bbox[axis2], g.h_bbox[axis2 + 1], g.h_bitwidth, ri);
}
// the quantised [min, max] of a face's corners along the three hash axes, 10 bits each (what in_quantised_box compares against)
PK_DEV unsigned long long pack_quantised_box(const DGrid& g, const double c0[4], const double c1[4], const double c2[4], bool three_axes) {
    unsigned long long b = 0;
    const double* cs[3] = {c0, c1, c2};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        unsigned lo = 0, hi = 1023;
        if (a < 2 || three_axes) {
            const double* c = cs[a];
            const double vlo = fmin(fmin(c[0], c[1]), fmin(c[2], c[3])), vhi = fmax(fmax(c[0], c[1]), fmax(c[2], c[3]));
            lo = quantize(vlo, g.h_bbox[2 * a], g.h_bbox[2 * a + 1], g.h_bitwidth, g.h_rinv[a]);
            hi = quantize(vhi, g.h_bbox[2 * a], g.h_bbox[2 * a + 1], g.h_bitwidth, g.h_rinv[a]);
        }
        b |= ((unsigned long long)(lo & 1023u) | ((unsigned long long)(hi & 1023u) << 10)) << (20 * a);
    }
    return b;
}
PK_DEV bool box_lists(const DGrid& g, unsigned long long b, double v0, double v1, double v2, bool three_axes) {
    const double v[3] = {v0, v1, v2};
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (a == 2 && !three_axes) break;
        const unsigned qv = quantize(v[a], g.h_bbox[2 * a], g.h_bbox[2 * a + 1], g.h_bitwidth, g.h_rinv[a]);
        const unsigned lo = (unsigned)(b >> (20 * a)) & 1023u, hi = (unsigned)(b >> (20 * a + 10)) & 1023u;
        ok = ok && lo <= qv && qv <= hi;
    }
    return ok;
}
PK_DEV uint32_t morton_code(const DGrid& g, const QPoint& q) {
    return (dilate_bits(quantize(q.qZ, g.h_bbox[4], g.h_bbox[5], g.h_bitwidth, g.h_rinv[2])) << 2) |
           (dilate_bits(quantize(q.qY, g.h_bbox[2], g.h_bbox[3], g.h_bitwidth, g.h_rinv[1])) << 1) |
           dilate_bits(quantize(q.qX, g.h_bbox[0], g.h_bbox[1], g.h_bitwidth, g.h_rinv[0]));
}

// curvilinear_point_in_cell (index_search.py:94-120); (yi, xi) must be a valid cell
// `listed` (optional): does the spatial-hash table list this face in the query's hash cell?  By construction of the table
// (spatialhash.py:269-387: a face is entered into every hash cell its quantised corner box overlaps) that is the case iff
// the query's quantised coordinates lie inside that box -- computed here from the corners already in registers.
PK_DEV bool point_in_cell(const DGrid& g, const QPoint& q, int yi, int xi, double& xsi, double& eta, const CellCache* cc = nullptr,
                          bool* listed = nullptr) {
    // node table rows: {lon, lat, X, Y, Z} of node (yi, xi) followed by node (yi, xi+1): 10 contiguous doubles
    const double* r0 = g.node_tab + ((int64_t)yi * g.nx + xi) * 5;
    const double* r1 = r0 + (int64_t)g.nx * 5;
    const bool use_cc = cc && cc->key;
    const int cell = yi * g.nx + xi;
    const bool hit = use_cc && cc->key[0] == cell;
    double* nd = use_cc ? cc->nodes : nullptr;
#define PK_ND(c_, m_) nd[((c_) * 2 + (m_)) * CC_LANES]
    if (g.spherical) {
        double eu[3], ev[3], pu[4], pv[4], xq, yq;
        // corner order c0=(yi,xi) c1=(yi,xi+1) c2=(yi+1,xi+1) c3=(yi+1,xi)
        if (hit && !listed) {  // the cell of the previous evaluation: its basis and projected corners are in the lane's cache
#pragma unroll
            for (int k = 0; k < 3; k++) { eu[k] = nd[(8 + k) * CC_LANES]; ev[k] = nd[(11 + k) * CC_LANES]; }
#pragma unroll
            for (int k = 0; k < 4; k++) { pu[k] = nd[(14 + k) * CC_LANES]; pv[k] = nd[(18 + k) * CC_LANES]; }
        } else if (g.cell_tab) {  // one contiguous record instead of two node rows + the two normalisations
            const double* ct = g.cell_tab + (int64_t)cell * CT_STRIDE;
            double boxd, unused;
            ldpair(ct + 8, eu[0], eu[1]); ldpair(ct + 10, eu[2], ev[0]); ldpair(ct + 12, ev[1], ev[2]);
            ldpair(ct + 14, pu[0], pu[1]); ldpair(ct + 16, pu[2], pu[3]); ldpair(ct + 18, pv[0], pv[1]); ldpair(ct + 20, pv[2], pv[3]);
            if (listed) {
                ldpair(ct + 22, boxd, unused);
                *listed = box_lists(g, (unsigned long long)__double_as_longlong(boxd), q.qX, q.qY, q.qZ, true);
            }
            if (use_cc && !hit) {
                double ll[8];
#pragma unroll
                for (int k = 0; k < 4; k++) ldpair(ct + 2 * k, ll[2 * k], ll[2 * k + 1]);
                cc->key[0] = -1;
#pragma unroll
                for (int k = 0; k < 8; k++) nd[k * CC_LANES] = ll[k];
#pragma unroll
                for (int k = 0; k < 3; k++) { nd[(8 + k) * CC_LANES] = eu[k]; nd[(11 + k) * CC_LANES] = ev[k]; }
#pragma unroll
                for (int k = 0; k < 4; k++) { nd[(14 + k) * CC_LANES] = pu[k]; nd[(18 + k) * CC_LANES] = pv[k]; }
            }
        } else {
            double cX[4], cY[4], cZ[4];
            double lon1, lat1, lon2, lat2;
            ldpair(r0 + 2, cX[0], cY[0]); ldpair(r0 + 4, cZ[0], lon1);
            ldpair(r0 + 6, lat1, cX[1]);  ldpair(r0 + 8, cY[1], cZ[1]);
            ldpair(r1 + 2, cX[3], cY[3]); ldpair(r1 + 4, cZ[3], lon2);
            ldpair(r1 + 6, lat2, cX[2]);  ldpair(r1 + 8, cY[2], cZ[2]);
            if (listed) *listed = in_quantised_box(g, cX, q.qX, 0) && in_quantised_box(g, cY, q.qY, 2) && in_quantised_box(g, cZ, q.qZ, 4);
            spherical_project_cell(cX, cY, cZ, eu, ev, pu, pv);
            if (use_cc && !hit) {  // the cached cell (the failed guess) is replaced; the key is set once the point is inside
                double lon0, lat0, lon3, lat3;
                ldpair(r0, lon0, lat0);
                ldpair(r1, lon3, lat3);
                cc->key[0] = -1;
                PK_ND(0, 0) = lon0; PK_ND(0, 1) = lat0; PK_ND(1, 0) = lon1; PK_ND(1, 1) = lat1;
                PK_ND(2, 0) = lon2; PK_ND(2, 1) = lat2; PK_ND(3, 0) = lon3; PK_ND(3, 1) = lat3;
#pragma unroll
                for (int k = 0; k < 3; k++) { nd[(8 + k) * CC_LANES] = eu[k]; nd[(11 + k) * CC_LANES] = ev[k]; }
#pragma unroll
                for (int k = 0; k < 4; k++) { nd[(14 + k) * CC_LANES] = pu[k]; nd[(18 + k) * CC_LANES] = pv[k]; }
            }
        }
        spherical_project_query(eu, ev, q, xq, yq);
        bilinear_inverse(pu, pv, xq, yq, xsi, eta);
    } else {
        double clon[4], clat[4];
        if (hit) {
#pragma unroll
            for (int k = 0; k < 4; k++) { clon[k] = PK_ND(k, 0); clat[k] = PK_ND(k, 1); }
        } else {
            if (g.cell_tab) {
                const double* ct = g.cell_tab + (int64_t)cell * CT_STRIDE;
#pragma unroll
                for (int k = 0; k < 4; k++) ldpair(ct + 2 * k, clon[k], clat[k]);
            } else {
                ldpair(r0, clon[0], clat[0]); ldpair(r0 + 5, clon[1], clat[1]);
                ldpair(r1, clon[3], clat[3]); ldpair(r1 + 5, clon[2], clat[2]);
            }
            if (use_cc) {
                cc->key[0] = -1;
#pragma unroll
                for (int k = 0; k < 4; k++) { PK_ND(k, 0) = clon[k]; PK_ND(k, 1) = clat[k]; }
            }
        }
        if (listed) *listed = in_quantised_box(g, clon, q.x, 0) && in_quantised_box(g, clat, q.y, 2);  // z is 0 for face and query
        bilinear_inverse(clon, clat, q.x, q.y, xsi, eta);
    }
#undef PK_ND
    const bool inside = (xsi >= 0) && (xsi <= 1) && (eta >= 0) && (eta <= 1);
    if (use_cc && !hit && inside) cc->key[0] = cell;
    return inside;
}

// _search_indices_curvilinear_2d (index_search.py:242-295) + SpatialHash.query (spatialhash.py:389-535) for one point:
// candidate -1 is the cell guessed from `ei` (if any), candidates 0.. are the faces of the query's hash cell in table
// order; the first candidate whose cell contains the point wins.  Hash hits return (xsi, eta) rounded to float32 like the
// reference's float32 coords_best buffer (spatialhash.py:505).  ONE point-in-cell call site serves both paths.
PK_DEV void curvilinear_search(const DGrid& g, double y, double x, bool use_guess, int gy, int gx, int& yi, int& xi, double& xsi,
                               double& eta, const CellCache* cc = nullptr, double* xsi_raw = nullptr, double* eta_raw = nullptr) {
    yi = GRID_SEARCH_ERROR;
    xi = GRID_SEARCH_ERROR;
    xsi = -1.0;
    eta = -1.0;
    const QPoint q = make_qpoint(g, y, x);
    const bool guess_ok = use_guess && gy >= 0 && gy < g.ny - 1 && gx >= 0 && gx < g.nx - 1;
    const uint32_t ncx = (uint32_t)(g.nx - 1);
    int64_t s = 0, c = -1;  // CSR range of the query's hash cell, looked up lazily (c < 0: not yet)
    auto lookup = [&]() {   // locate the query's Morton code in the key table (spatialhash.py:430-470)
        c = 0;
        if (isfinite(x) && isfinite(y) && g.h_nkeys > 0) {
            const uint32_t code = morton_code(g, q);
            int64_t lo = 0, hi = g.h_nkeys;
            if (g.h_dir) {  // keys with the same top bits: a handful instead of all of them
                const uint32_t b = code >> g.h_dir_shift;
                lo = g.h_dir[b];
                hi = g.h_dir[b + 1];
            }
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (g.h_keys[mid] < code) lo = mid + 1; else hi = mid;
            }
            if (lo < g.h_nkeys && g.h_keys[lo] == code) { s = g.h_starts[lo]; c = g.h_counts[lo]; }
        }
    };
    for (int64_t k = guess_ok ? -1 : 0;; k++) {
        int j, i;
        if (k < 0) {
            j = gy;
            i = gx;
        } else {
            if (c < 0) lookup();
            if (k >= c) return;
            const uint32_t face = g.h_faces[s + k];
            j = (int)(face / ncx);
            i = (int)(face % ncx);
        }
        double xs, et;
        if (point_in_cell(g, q, j, i, xs, et, cc)) {
            yi = j;
            xi = i;
            xsi = k < 0 ? xs : (double)(float)xs;
            eta = k < 0 ? et : (double)(float)et;
            if (xsi_raw) { *xsi_raw = xs; *eta_raw = et; }
            return;
        }
        if (k < 0 && g.walk_ok) {
            // The particle left the guessed cell.  On a mesh without coincident nodes (cells cannot overlap) at most one cell
            // holds it, almost always the neighbour the barycentric coordinates point at: test that one before walking the
            // ~10-20 faces of the hash cell in table order (one point-in-cell test + two dependent loads each).  Accepted only
            // (a) well inside -- no tie with an adjacent cell, which the table order would have to break -- and (b) if that
            // face is listed in the query's hash cell, because a face the table does not list there is one the reference
            // cannot find (it answers GRID_SEARCH_ERROR: the xyz box of the four corners does not cover the whole curved
            // cell).  The coordinates are rounded like a hash hit (spatialhash.py:505).
            const int dj = et < 0 ? -1 : (et > 1 ? 1 : 0), di = xs < 0 ? -1 : (xs > 1 ? 1 : 0);
            const int nj = gy + dj, ni = gx + di;
            if ((dj | di) != 0 && nj >= 0 && nj < g.ny - 1 && ni >= 0 && ni < g.nx - 1) {
                double xs2, et2;
                const double m = 1e-9;
                bool listed = false;
                if (point_in_cell(g, q, nj, ni, xs2, et2, cc, &listed) && xs2 > m && xs2 < 1 - m && et2 > m && et2 < 1 - m) {
                    if (listed) {
                        yi = nj;
                        xi = ni;
                        xsi = (double)(float)xs2;
                        eta = (double)(float)et2;
                        if (xsi_raw) { *xsi_raw = xs2; *eta_raw = et2; }
                        return;
                    }
                }
            }
        }
    }
}

// ravel/unravel of `ei` over XGrid.axes (basegrid.py:83-152, 219-278)
PK_DEV int64_t ravel_ei(const DGrid& g, int zi, int yi, int xi) {
    int64_t ei = 0, stride = 1;
    if (g.has_x) { ei += (int64_t)xi * stride; stride *= g.xdim; }
    if (g.has_y) { ei += (int64_t)yi * stride; stride *= g.ydim; }
    if (g.has_z) { ei += (int64_t)zi * stride; }
    return ei;
}
PK_DEV int64_t floordiv64(int64_t a, int64_t b) {
    if (b == 0) return 0;
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
    return q;
}
PK_DEV int64_t mod64(int64_t a, int64_t b) {
    if (b == 0) return 0;
    int64_t r = a % b;
    if (r != 0 && ((r < 0) != (b < 0))) r += b;
    return r;
}
PK_DEV void unravel_yx(const DGrid& g, int64_t ei, int& yi, int& xi) {
    // strides[i] = prod(dims[i:]); idx[i] = ei // strides[i+1]; ei %= strides[i+1]
    int64_t sx = g.has_x ? (int64_t)g.xdim : 1;             // stride of Y (product of dims after Y)
    int64_t sy = (g.has_y ? (int64_t)g.ydim : 1) * sx;      // stride of Z
    if (g.has_z && (g.has_y || g.has_x)) ei = mod64(ei, sy);
    if (g.has_y) {
        if (g.has_x) { yi = (int)floordiv64(ei, sx); ei = mod64(ei, sx); }
        else yi = (int)ei;
    } else yi = 0;
    xi = g.has_x ? (int)ei : 0;
}

// ---- per-particle evaluation context -------------------------------------------------------------------
struct PCtx {
    int state;
    bool pf;             // particle positions are stored as float32 (default Particle, particle.py:123-178)
    int hz, hy, hx, ht;  // search hints of the main grid (indices of the previous evaluation)
    bool hyx_valid;      // (hy, hx) is an in-range cell whose ravelled index is the particle's ei on the main grid
    unsigned first_eval;  // bit g: no evaluation on grid g yet in this execute() call
    int32_t ei0, ei1, ei2, ei3;  // the particle's `ei` row, one register per grid (no dynamic indexing -> no scratch)
    bool u32, v32;  // the u / v ARRAYS of the last eval_uvw are float32 in the reference (AdvectionRK45's stage-1 products)
    bool oob;       // some sample of this context was masked to 0 by _mask_outofbounds_values (field.py:359-370); read by pk_eval only
    int64_t row;    // device row of the particle (kernels that write user Variables)
    bool zpos_f32;  // the z of the NEXT sample is a float32 particle column (set by the caller next to the sample's pos_f32, which speaks for
                    // y and x): the 2-D kernels hand `particles.z` to every stage unchanged -- on a float32 depth axis its zeta stays a
                    // float32 array (index_search.py:51) while the stage positions y1, x1 are float64
    unsigned it;    // 1-based iteration of the loop of kernel.py:190 the particle is in (0: outside it) ...
    int klo;        // ... and kernel slot * 1000 + samples taken so far in this iteration's call(s) of that kernel: the key of the next sample (twe_listed)
};

PK_DEV int32_t ei_get(const PCtx& c, int g) { return g == 0 ? c.ei0 : (g == 1 ? c.ei1 : (g == 2 ? c.ei2 : c.ei3)); }
PK_DEV void ei_set(PCtx& c, int g, int32_t v) {  // value selects only: PCtx must stay in registers
    c.ei0 = g == 0 ? v : c.ei0;
    c.ei1 = g == 1 ? v : c.ei1;
    c.ei2 = g == 2 ? v : c.ei2;
    c.ei3 = g == 3 ? v : c.ei3;
}
PK_DEV bool take_first_eval(PCtx& c, int g) {
    const bool f = (c.first_eval >> g) & 1u;
    c.first_eval &= ~(1u << g);
    return f;
}

// what follows the search proper: the ravelled `ei` (xgrid.py:349-356), the hints of the next search on the main grid, and the
// state update of field.py:307-356
PK_DEV void grid_search_finish(const DGrid& g, bool hint, bool curv, int32_t* ei, PCtx& c, const GPos& p) {
    const int64_t rav = ravel_ei(g, p.zi, p.yi, p.xi);
    *ei = (int32_t)rav;
    if (hint) {
        c.hz = p.zi; c.hy = p.yi; c.hx = p.xi;
        c.hyx_valid = curv && p.zi >= 0 && p.yi >= 0 && p.xi >= 0 && rav == (int64_t)*ei;
    }
    int s = c.state;
    if (p.xi == -1 && s < PK_ERROROUTOFBOUNDS) s = PK_ERROROUTOFBOUNDS;
    if (p.xi == GRID_SEARCH_ERROR && s < PK_ERRORGRIDSEARCHING) s = PK_ERRORGRIDSEARCHING;
    if (p.yi == -1 && s < PK_ERROROUTOFBOUNDS) s = PK_ERROROUTOFBOUNDS;
    if (p.yi == GRID_SEARCH_ERROR && s < PK_ERRORGRIDSEARCHING) s = PK_ERRORGRIDSEARCHING;
    if (p.zi == RIGHT_OUT_OF_BOUNDS && s < PK_ERROROUTOFBOUNDS) s = PK_ERROROUTOFBOUNDS;
    if (p.zi == LEFT_OUT_OF_BOUNDS && s < PK_ERRORTHROUGHSURFACE) s = PK_ERRORTHROUGHSURFACE;
    c.state = s;
}

// The grid position the velocity sample of a kernel found, kept for the scalar samples the same kernel takes at the SAME point of
// the SAME grid (AdvectionDiffusionM1 / EM read Kh_zonal and Kh_meridional at the particle position after UV,
// _advectiondiffusion.py:44-62).  The reference searches again, starting from the cell the previous sample left in `ei`.  On a
// rectilinear grid the result is a pure function of the point.  On a curvilinear grid a search that STARTS in the cell containing
// the point returns that cell with the float64 (xsi, eta) of the point-in-cell test (index_search.py:269-285) -- GPos::xsi_raw /
// eta_raw, whichever way the memo's own search found the cell -- while one that starts elsewhere goes through the spatial hash
// (float32-rounded coordinates, and only faces the table lists).  So the memo is re-used exactly when the sample before this one left
// the particle's `ei` in the memo's cell, which eval_scalar reads off the search hints: always for the Kh sample that directly follows
// UV, for the one after Kh(y - dres) only if that point lies in the same cell (a 2500-seed fuzz run pins this; a memo re-used
// regardless differs from the reference in the float32 rounding of (xsi, eta)).  Other lanes search again.
struct SearchMemo {
    GPos p;
    int grid;  // -1: nothing to re-use
};

// XGrid.search (xgrid.py:316-356) + ei write + state update (field.py:307-356)
// KIND: 0 rectilinear, 1 curvilinear, -1 decide at run time (scalar fields on secondary grids)
// TYPED: some grid of the fieldset stores a coordinate as float32, so NumPy's float32 arithmetic on coordinate / barycentric
// arrays has to be reproduced (GPos::x32 / e32 / z32, the f32 paths of search_1d).  false: those are compile-time constants
// (only the float32 (xsi, eta) of an unguessed curvilinear evaluation, GPos::w32, remains a run-time property) and the
// programs built for float64 coordinates carry none of the emulation.
template <int KIND, bool TYPED>
PK_DEV void grid_search(const DGrid& g, const Coords* mc, double z, double y, double x, bool pos_f32, int32_t* ei,
                        PCtx& c, bool use_guess, GPos& p) {
    const bool curv = (KIND < 0) ? (g.kind == 1) : (KIND == 1);
    const double* depth = mc ? mc->depth : g.depth;
    const double* lat = mc ? mc->lat : g.lat;
    const double* lon = mc ? mc->lon : g.lon;
    const bool hint = mc != nullptr;
    const bool zf32 = TYPED && g.depth_f32, yf32 = TYPED && g.lat_f32, xf32 = TYPED && g.lon_f32;
    const bool zpos_f32 = TYPED && c.zpos_f32;
    if (g.has_z) search_1d(depth, g.nz, mc ? mc->z0 : g.depth[0], mc ? mc->z1 : g.depth[g.nz - 1], z, zf32, zpos_f32, hint ? c.hz : 0, p.zi, p.zeta);
    else { p.zi = 0; p.zeta = 0.0; }
    p.w32 = curv && !use_guess;
    p.z32 = g.has_z && zf32 && zpos_f32 && g.nz >= 2;
    p.x32 = curv ? p.w32 : (g.has_x && xf32 && pos_f32 && g.nx >= 2);
    p.e32 = curv ? p.w32 : (g.has_y && yf32 && pos_f32 && g.ny >= 2);
    if (curv) {
        int gy = 0, gx = 0;
        if (use_guess) {  // index_search.py:269-274
            // unravel(ei) of the cell the previous evaluation found is that cell: skip the 64-bit div/mod chain
            if (hint && c.hyx_valid) { gy = c.hy; gx = c.hx; }
            else unravel_yx(g, (int64_t)*ei, gy, gx);
        }
        curvilinear_search(g, y, x, use_guess, gy, gx, p.yi, p.xi, p.xsi, p.eta, mc ? &mc->cc : nullptr, &p.xsi_raw, &p.eta_raw);
    } else {
        if (g.has_y) search_1d(lat, g.ny, mc ? mc->y0 : g.lat[0], mc ? mc->y1 : g.lat[g.ny - 1], y, yf32, pos_f32, hint ? c.hy : 0, p.yi, p.eta);
        else { p.yi = 0; p.eta = 0.0; }
        if (g.has_x) search_1d(lon, g.nx, mc ? mc->x0 : g.lon[0], mc ? mc->x1 : g.lon[g.nx - 1], x, xf32, pos_f32, hint ? c.hx : 0, p.xi, p.xsi);
        else { p.xi = 0; p.xsi = 0.0; }
    }
    grid_search_finish(g, hint, curv, ei, c, p);
}

// _search_time_index (index_search.py:65-91). false => OutsideTimeInterval
PK_DEV bool time_search(const DField& f, const double* time, double t, int hint, GPos& p) {
    if (!f.has_time_interval) { p.ti = 0; p.tau = 0.0; return true; }
    if (!(0 <= t) || !(t <= f.tlen)) return false;  // utils/time.py:60-62
    search_1d(time, f.nt, f.tfirst, f.tlast, t, false, false, hint, p.ti, p.tau);
    return true;
}

// ---- corner gather + interpolation -----------------------------------------------------------------------
template <class FT>
PK_DEV double ldv(const FT* p, int64_t off) { return (double)p[off]; }

// element offset of time level `ti` inside the field's (possibly component-packed) buffer, including the component
PK_DEV int64_t slot_off(const DField& f, int ti) {
    int s = (f.nslots >= f.nt) ? ti : (ti % f.nslots);
    return (int64_t)s * f.st_t * f.ncomp + f.comp;
}

// Element offsets of the 16 bracketing corners (_gather_corners / _get_corner_data_Agrid, _xinterpolators.py:25-96),
// computed once per evaluation and shared by every field with the same array layout (U, V, W of an A-grid).
struct Corners {
    int64_t ot0, ot1;   // slot offsets of the two time levels
    int o[2][2];        // [z][y] in-level offset of the x0 corner (a level holds < 2^31 elements)
    int dx;             // offset of the x1 corner relative to x0 (1, or 0 when clipped / no x dimension)
    bool lenT, lenZ;
    bool pairs;         // uniform: x-corners are adjacent elements -> one wide load per corner pair
};
PK_DEV Corners make_corners(const DField& f, const GPos& p) {
    Corners k;
    k.lenT = p.tau > 0;
    k.lenZ = !(p.zeta <= 0);  // also for a NaN zeta: it must poison the value like in a batch whose lenZ is 2
    k.ot0 = slot_off(f, p.ti);
    k.ot1 = slot_off(f, mini(p.ti + 1, f.nt - 1));
    const int nc = f.ncomp;
    const int sz = (int)f.st_z * nc, sy = (int)f.st_y * nc, sx = (int)f.st_x * nc;
    const int oz0 = p.zi * sz, oz1 = mini(p.zi + 1, f.nz - 1) * sz;
    const int oy0 = p.yi * sy, oy1 = mini(p.yi + 1, f.ny - 1) * sy;
    const int ox0 = p.xi * sx;
    k.dx = mini(p.xi + 1, f.nx - 1) * sx - ox0;
    k.pairs = (f.st_x == 1) && (f.nx >= 2) && (nc == 1);  // then xi <= nx-2 for every in-bounds lane, so x1 == x0 + 1
    k.o[0][0] = oz0 + oy0 + ox0;
    k.o[0][1] = oz0 + oy1 + ox0;
    k.o[1][0] = oz1 + oy0 + ox0;
    k.o[1][1] = oz1 + oy1 + ox0;
    return k;
}
PK_DEV bool same_layout(const DField& a, const DField& b) {
    return a.ncomp == 1 && b.ncomp == 1 && a.st_t == b.st_t && a.st_z == b.st_z && a.st_y == b.st_y && a.st_x == b.st_x && a.nt == b.nt && a.nz == b.nz &&
           a.ny == b.ny && a.nx == b.nx && a.nslots == b.nslots;
}
template <class FT>
PK_DEV void ld2(const FT* p, int off, int dx, bool pairs, double& a, double& b) {
    if (pairs) {
        ldpair(p + off, a, b);
    } else {  // a field without an x dimension (or of extent 1)
        a = (double)p[off];
        b = (double)p[off + dx];
    }
}

// One depth level of XLinear: the 4 (y, x) corners at both time levels -> time-interpolated corner values.
template <class FT>
PK_DEV void xlinear_level(const FT* d0, const FT* d1, const Corners& k, int iz, double tau, double c[2][2]) {
    double a[2][2], b[2][2];
    ld2(d0, k.o[iz][0], k.dx, k.pairs, a[0][0], a[0][1]);
    ld2(d0, k.o[iz][1], k.dx, k.pairs, a[1][0], a[1][1]);
    if (k.lenT) {
        ld2(d1, k.o[iz][0], k.dx, k.pairs, b[0][0], b[0][1]);
        ld2(d1, k.o[iz][1], k.dx, k.pairs, b[1][0], b[1][1]);
    }
#pragma unroll
    for (int iy = 0; iy < 2; iy++)
#pragma unroll
        for (int ix = 0; ix < 2; ix++) c[iy][ix] = k.lenT ? a[iy][ix] * (1 - tau) + b[iy][ix] * tau : a[iy][ix];
}

// XLinear.interp (_xinterpolators.py:112-153): lerp in t, then z, then bilinear in (eta, xsi).
// The gather is issued in two batches (depth level z0, then z1) of up to 4 wide loads each: all 8 (16 scalar) loads at
// once would hold ~64 VGPRs of data + addresses per field and cap the kernel at 2 waves/SIMD.
template <class FT>
PK_DEV double xlinear(const DField& f, const Corners& k, const GPos& p, bool* out32 = nullptr) {
    const FT* d0 = (const FT*)f.data + k.ot0;
    const FT* d1 = (const FT*)f.data + k.ot1;
    const double tau = p.tau, zeta = p.zeta, xsi = p.xsi, eta = p.eta;
    double c[2][2];
    xlinear_level<FT>(d0, d1, k, 0, tau, c);
    const bool typed = p.x32 || p.e32 || p.z32;  // some bcoord array is float32 in the reference (uniform across the wave)
    const bool c32 = sizeof(FT) == 4 && !k.lenT;  // corner values are float32 data untouched by the float64 time lerp
    if (k.lenZ) {
        PK_FIELD_FENCE();
        double c1[2][2];
        xlinear_level<FT>(d0, d1, k, 1, tau, c1);
        if (typed) {
            const TV z = TV{zeta, p.z32}, omz = tv_one_minus(z);
#pragma unroll
            for (int iy = 0; iy < 2; iy++)
#pragma unroll
                for (int ix = 0; ix < 2; ix++) c[iy][ix] = tv_add(tv_mul(TV{c[iy][ix], c32}, omz), tv_mul(TV{c1[iy][ix], c32}, z)).v;
        } else {
#pragma unroll
            for (int iy = 0; iy < 2; iy++)
#pragma unroll
                for (int ix = 0; ix < 2; ix++) c[iy][ix] = c[iy][ix] * (1 - zeta) + c1[iy][ix] * zeta;
        }
    }
    if (typed) {  // _xinterpolators.py:146-151 with NumPy's dtype propagation
        const bool cz32 = c32 && (!k.lenZ || p.z32);
        const TV x = TV{xsi, p.x32}, e = TV{eta, p.e32}, omx = tv_one_minus(x), ome = tv_one_minus(e);
        TV r = tv_mul(tv_mul(omx, ome), TV{c[0][0], cz32});
        r = tv_add(r, tv_mul(tv_mul(x, ome), TV{c[0][1], cz32}));
        r = tv_add(r, tv_mul(tv_mul(omx, e), TV{c[1][0], cz32}));
        r = tv_add(r, tv_mul(tv_mul(x, e), TV{c[1][1], cz32}));
        if (out32) *out32 = r.f32;
        return r.v;
    }
    if (out32) *out32 = false;
    return (1 - xsi) * (1 - eta) * c[0][0] + xsi * (1 - eta) * c[0][1] + (1 - xsi) * eta * c[1][0] + xsi * eta * c[1][1];
}
template <class FT>
PK_DEV double xlinear(const DField& f, const GPos& p) { return xlinear<FT>(f, make_corners(f, p), p); }

// _Spatialslip (_xinterpolators.py:386-477): XFreeslip (a = 1, b = 0) and XPartialslip (a = b = 0.5).  A corner is "land"
// when np.isclose(U, 0) & np.isclose(V, 0) at the first bracketing time level; a cell row / column that is all land scales
// the tangential velocity by (a + b*eta)/eta etc.  lenZ (does the z1 level take part in the u / v land tests) is batch-global
// in the reference, `2 if np.any(zeta > 0) else 1`: here per particle (zeta > 0) unless `force_lenz` (1 / 2) gives the value of
// the batch (pk_eval: one call is one batch); the W factors always use both depth levels like the reference.
// NumPy dtypes: f_u = ones_like(xsi), f_v = ones_like(eta), f_w = ones_like(zeta) take the dtype of THAT barycentric array, the
// factor expressions the dtype of the coordinate they are made of (product first, then the quotient), `f[land] = ...` casts
// back to f's dtype, and the in-place `u /= ...`, `v /= ...` keep the dtype XLinear returned.
template <class FT>
PK_DEV void slip_velocity(const DGrid& g, const DField& U, const DField& V, const DField* W, const GPos& p, double ypos,
                          bool ypos_f32, double a_, double b_, int force_lenz, double& u, double& v, double& w, bool& u32o, bool& v32o) {
    const Corners ku = make_corners(U, p);
    const bool same_v = same_layout(U, V);
    const Corners kv = same_v ? ku : make_corners(V, p);
    bool u32 = false, v32 = false, w32 = false;
    NV uu = nvv(xlinear<FT>(U, ku, p, &u32), 0);
    NV vv = nvv(xlinear<FT>(V, kv, p, &v32), 0);
    NV ww = nvv(0.0, 0);
    if (W) ww.v = same_layout(U, *W) ? xlinear<FT>(*W, ku, p, &w32) : xlinear<FT>(*W, make_corners(*W, p), p, &w32);
    uu.dt = u32; vv.dt = v32; ww.dt = w32;
    const double zero_tol = sizeof(FT) == 4 ? (double)(float)1e-8 : 1e-8;  // np.isclose(v, 0.0): |v| <= atol
    const FT* du = (const FT*)U.data + ku.ot0;
    const FT* dv = (const FT*)V.data + kv.ot0;
    bool land[2][2][2];
#pragma unroll
    for (int iz = 0; iz < 2; iz++)
#pragma unroll
        for (int iy = 0; iy < 2; iy++) {
            double u0, u1, v0, v1;
            ld2(du, ku.o[iz][iy], ku.dx, ku.pairs, u0, u1);
            ld2(dv, kv.o[iz][iy], kv.dx, kv.pairs, v0, v1);
            land[iz][iy][0] = fabs(u0) <= zero_tol && fabs(v0) <= zero_tol;
            land[iz][iy][1] = fabs(u1) <= zero_tol && fabs(v1) <= zero_tol;
        }
    const NV xsi = nvv(p.xsi, p.x32), eta = nvv(p.eta, p.e32);
    const NV one = nvv(1.0, 2), na = nvv(a_, 2), nb = nvv(b_, 2);
    const bool z2 = force_lenz ? force_lenz == 2 : p.zeta > 0;
    const bool row0 = land[0][0][0] && land[0][0][1], row0z = land[1][0][0] && land[1][0][1];
    const bool row1 = land[0][1][0] && land[0][1][1], row1z = land[1][1][0] && land[1][1][1];
    const bool col0 = land[0][0][0] && land[0][1][0], col0z = land[1][0][0] && land[1][1][0];
    const bool col1 = land[0][0][1] && land[0][1][1], col1z = land[1][0][1] && land[1][1][1];
    // f[land] = f[land] * (a + b*c) / c   and   f[land] * (1 - b*c) / (1 - c)
    auto fac_lo = [&](NV f, NV c) { return nv_cast(nv_div(nv_mul(f, nv_add(na, nv_mul(nb, c))), c), f.dt); };
    auto fac_hi = [&](NV f, NV c) { return nv_cast(nv_div(nv_mul(f, nv_sub(one, nv_mul(nb, c))), nv_sub(one, c)), f.dt); };
    NV f_u = nvv(1.0, p.x32);
    if (row0 && (!z2 || row0z) && p.eta > 0) f_u = fac_lo(f_u, eta);
    if (row1 && (!z2 || row1z) && p.eta < 1) f_u = fac_hi(f_u, eta);
    uu = nv_mul(uu, f_u);
    if (g.spherical) {  // the reference hard-codes 1852*60 here (== deg2m of the default sphere)
        const NV conv = ypos_f32 ? NV{(double)(111120.0f * cosf((float)ypos * DEG2RADF)), 1} : NV{111120 * cos_lat(ypos * DEG2RAD), 0};
        uu = nv_cast(nv_div(uu, conv), uu.dt);
    }
    NV f_v = nvv(1.0, p.e32);
    if (col0 && (!z2 || col0z) && p.xsi > 0) f_v = fac_lo(f_v, xsi);
    if (col1 && (!z2 || col1z) && p.xsi < 1) f_v = fac_hi(f_v, xsi);
    vv = nv_mul(vv, f_v);
    if (g.spherical) vv = nv_cast(nv_div(vv, nvv(111120.0, 2)), vv.dt);
    if (W) {
        NV f_w = nvv(1.0, p.z32);
        if (row0 && row0z && p.eta > 0) f_w = fac_lo(f_w, eta);
        if (row1 && row1z && p.eta < 1) f_w = fac_hi(f_w, eta);
        if (col0 && col0z && p.xsi > 0) f_w = fac_lo(f_w, xsi);
        if (col1 && col1z && p.xsi < 1) f_w = fac_hi(f_w, xsi);
        ww = nv_mul(ww, f_w);
    }
    u = uu.v;
    v = vv.v;
    w = ww.v;
    u32o = uu.dt == 1;
    v32o = vv.dt == 1;
}

// _geodetic_distance (utils/interpolation.py:178-185), including NumPy's float32 behaviour for f32 coordinates
PK_DEV double geodetic_distance(const DGrid& g, double lat1, double lat2, double lon1, double lon2, double lat, bool cf32) {
    if (g.spherical) {
        if (cf32) {
            double dl = (double)(((float)lon2 - (float)lon1) * (float)g.deg2m);
            float dlaf = ((float)lat2 - (float)lat1) * (float)g.deg2m;
            double a = dl * cos_lat(DEG2RAD * lat);
            return sqrt(a * a + (double)(dlaf * dlaf));
        }
        double dl = (lon2 - lon1) * g.deg2m, dla = (lat2 - lat1) * g.deg2m;
        double a = dl * cos_lat(DEG2RAD * lat);
        return sqrt(a * a + dla * dla);
    }
    if (cf32) {
        float a = (float)lon2 - (float)lon1, b = (float)lat2 - (float)lat1;
        return (double)sqrtf(a * a + b * b);
    }
    double a = lon2 - lon1, b = lat2 - lat1;
    return sqrt(a * a + b * b);
}

// two bracketing face values, reduced over time (_xinterpolators.py:249-270)
PK_DEV int64_t cgrid_off(const DField& f, int z, int y, int x) {
    return ((int64_t)z * f.st_z + (int64_t)y * f.st_y + (int64_t)x * f.st_x) * f.ncomp;
}

PK_DEV double pymod360(double v) {  // Python/NumPy % with a positive divisor
    if (v >= 0.0 && v < 360.0) return v;  // fmod is exact, so this IS its result on the common path (no 50-instruction fmod)
    double q = fmod(v, 360.0);
    if (q < 0) q += 360.0;
    return q;
}
PK_DEV float pymod360f(float v) {
    if (v >= 0.0f && v < 360.0f) return v;
    float q = fmodf(v, 360.0f);
    if (q < 0) q += 360.0f;
    return q;
}

// np.column_stack of four entries followed by einsum("ij,ji->i", ., q) (phi2D_lin and the Jacobian rows, utils/interpolation.py:
// 25-31,188-198): the entries are cast to their common dtype, products and running sum take the promoted dtype of both operands.
// float64 entries against float32 coordinates run through einsum's buffered two-accumulator loop, (p0 + p2) + (p1 + p3); every
// other combination adds left to right (batches of two or more particles; measured on NumPy 2.2).
PK_DEV NV nv_dot4(const NV e[4], const NV q[4]) {
    int dt = 2;
#pragma unroll
    for (int k = 0; k < 4; k++) dt = nv_res(dt, e[k].dt);
    if (dt == 2) dt = 0;
    const int odt = nv_res(dt, q[0].dt);
    if (dt == 0 && q[0].dt == 1) {
        const double p0 = e[0].v * q[0].v, p1 = e[1].v * q[1].v, p2 = e[2].v * q[2].v, p3 = e[3].v * q[3].v;
        return NV{(p0 + p2) + (p1 + p3), odt};
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double a = dt == 1 ? (double)(float)e[k].v : e[k].v;
        if (odt == 1) acc = (double)((float)acc + (float)a * (float)q[k].v);
        else acc = acc + a * q[k].v;
    }
    return NV{acc, odt};
}
PK_DEV void nv_phi2d(NV eta, NV xsi, NV out[4]) {  // utils/interpolation.py:25-31
    const NV one = nvv(1.0, 2);
    const NV omx = nv_sub(one, xsi), ome = nv_sub(one, eta);
    out[0] = nv_mul(omx, ome);
    out[1] = nv_mul(xsi, ome);
    out[2] = nv_mul(xsi, eta);
    out[3] = nv_mul(omx, eta);
}
PK_DEV NV nv_cos_deg(NV lat) {  // np.cos(np.pi / 180.0 * lat)
    const NV r = nv_mul(nvv(3.14159265358979323846 / 180.0, 2), lat);
    return r.dt == 1 ? NV{(double)cosf((float)r.v), 1} : NV{cos_lat(r.v), r.dt};
}
PK_DEV NV nv_geodetic(const DGrid& g, NV lat1, NV lat2, NV lon1, NV lon2, NV lat) {  // utils/interpolation.py:178-185
    if (g.spherical) {
        const NV a = nv_mul(nv_mul(nv_sub(lon2, lon1), nvv(g.deg2m, 2)), nv_cos_deg(lat));
        const NV b = nv_mul(nv_sub(lat2, lat1), nvv(g.deg2m, 2));
        return nv_sqrt(nv_add(nv_mul(a, a), nv_mul(b, b)));
    }
    const NV a = nv_sub(lon2, lon1), b = nv_sub(lat2, lat1);
    return nv_sqrt(nv_add(nv_mul(a, a), nv_mul(b, b)));
}
// two bracketing face values reduced over time, with their dtype (_xinterpolators.py:249-270): tau is a float64 array
template <class FT>
PK_DEV void cgrid_pair_t(const DField& f, const GPos& p, int64_t offA, int64_t offB, NV out[2]) {
    const FT* d = (const FT*)f.data;
    const int64_t s0 = slot_off(f, p.ti);
    double a = ldv(d, s0 + offA), b = ldv(d, s0 + offB);
    int dt = sizeof(FT) == 4 ? 1 : 0;
    if (p.tau > 0) {
        const int64_t s1 = slot_off(f, mini(p.ti + 1, f.nt - 1));
        a = a * (1 - p.tau) + ldv(d, s1 + offA) * p.tau;
        b = b * (1 - p.tau) + ldv(d, s1 + offB) * p.tau;
        dt = 0;
    }
    out[0] = nvv(a, dt);
    out[1] = nvv(b, dt);
}
// CGrid_Velocity.interp with every operation carrying its NumPy dtype: float32 coordinate arrays (lon_f32 / lat_f32), float32
// barycentric arrays (GPos::x32 / e32 / z32) and float32 field data that no float64 time lerp has touched.  Used by the programs
// built for fieldsets with float32 coordinates (TYPED); pxv / pyv: corner coordinates after the antimeridian unwrapping.
template <class FT>
PK_DEV void cgrid_velocity_typed(const DGrid& g, const DField& U, const DField& V, const DField* W, const GPos& p, const double pxv[4],
                                 const double pyv[4], double ypos, bool ypos_f32, double& u, double& v, double& w, bool& u32, bool& v32) {
    const int xi = p.xi, yi = p.yi, zi = p.zi;
    const int ydim = U.ny, xdim = U.nx, zdim = U.nz;
    const NV xsi = nvv(p.xsi, p.x32), eta = nvv(p.eta, p.e32), zeta = nvv(p.zeta, p.z32);
    const NV one = nvv(1.0, 2), zero = nvv(0.0, 2);
    NV px[4], py[4], phi[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { px[k] = nvv(pxv[k], g.lon_f32 ? 1 : 0); py[k] = nvv(pyv[k], g.lat_f32 ? 1 : 0); }
    nv_phi2d(zero, xsi, phi);
    const NV c1 = nv_geodetic(g, py[0], py[1], px[0], px[1], nv_dot4(phi, py));
    nv_phi2d(eta, one, phi);
    const NV c2 = nv_geodetic(g, py[1], py[2], px[1], px[2], nv_dot4(phi, py));
    nv_phi2d(one, xsi, phi);
    const NV c3 = nv_geodetic(g, py[2], py[3], px[2], px[3], nv_dot4(phi, py));
    nv_phi2d(eta, zero, phi);
    const NV c4 = nv_geodetic(g, py[3], py[0], px[3], px[0], nv_dot4(phi, py));
    NV cd[2];
    const int yi_o = clampi(yi + g.off_y, 0, ydim - 1), xi_1 = clampi(xi + 1, 0, xdim - 1);
    cgrid_pair_t<FT>(U, p, cgrid_off(U, zi, yi_o, xi), cgrid_off(U, zi, yi_o, xi_1), cd);
    const NV U0 = nv_mul(cd[0], c4), U1 = nv_mul(cd[1], c2);
    const NV omx = nv_sub(one, xsi), ome = nv_sub(one, eta);
    const NV Uvel = nv_add(nv_mul(omx, U0), nv_mul(xsi, U1));
    const int yi_1 = clampi(yi + 1, 0, ydim - 1), xi_o = clampi(xi + g.off_x, 0, xdim - 1);
    cgrid_pair_t<FT>(V, p, cgrid_off(V, zi, yi, xi_o), cgrid_off(V, zi, yi_1, xi_o), cd);
    const NV V0 = nv_mul(cd[0], c1), V1 = nv_mul(cd[1], c3);
    const NV Vvel = nv_add(nv_mul(ome, V0), nv_mul(eta, V1));
    // _compute_jacobian_determinant (utils/interpolation.py:188-198)
    const NV dxs[4] = {nv_sub(eta, one), ome, eta, nv_neg(eta)};
    const NV det[4] = {nv_sub(xsi, one), nv_neg(xsi), xsi, omx};
    const NV dxdxsi = nv_dot4(dxs, px), dxdeta = nv_dot4(det, px), dydxsi = nv_dot4(dxs, py), dydeta = nv_dot4(det, py);
    NV jac = nv_sub(nv_mul(dxdxsi, dydeta), nv_mul(dxdeta, dydxsi));
    if (g.spherical) jac = nv_mul(jac, nvv(g.deg2m, 2));
    const NV A = nv_sub(nv_mul(nv_neg(ome), Uvel), nv_mul(omx, Vvel));
    const NV B = nv_sub(nv_mul(ome, Uvel), nv_mul(xsi, Vvel));
    const NV C = nv_add(nv_mul(eta, Uvel), nv_mul(xsi, Vvel));
    const NV D = nv_add(nv_mul(nv_neg(eta), Uvel), nv_mul(omx, Vvel));
    NV uu = nv_div(nv_add(nv_add(nv_add(nv_mul(A, px[0]), nv_mul(B, px[1])), nv_mul(C, px[2])), nv_mul(D, px[3])), jac);
    NV vv = nv_div(nv_add(nv_add(nv_add(nv_mul(A, py[0]), nv_mul(B, py[1])), nv_mul(C, py[2])), nv_mul(D, py[3])), jac);
    if (g.spherical) {  // :311-314, in place: u and v keep their dtype
        const NV conv = nv_mul(nvv(g.deg2m, 2), ypos_f32 ? NV{(double)cosf((float)ypos * DEG2RADF), 1} : NV{cos_lat(ypos * DEG2RAD), 0});
        uu = nv_cast(nv_div(uu, conv), uu.dt);
        vv = nv_cast(nv_div(vv, conv), vv.dt);
    }
    u = uu.v;
    v = vv.v;
    u32 = uu.dt == 1;
    v32 = vv.dt == 1;
    if (W) {  // :316-328
        const int zi_0 = clampi(zi + g.off_z, 0, zdim - 1), zi_1 = clampi(zi + g.off_z + 1, 0, zdim - 1);
        cgrid_pair_t<FT>(*W, p, cgrid_off(*W, zi_0, yi_o, xi_o), cgrid_off(*W, zi_1, yi_o, xi_o), cd);
        w = nv_add(nv_mul(cd[0], nv_sub(one, zeta)), nv_mul(cd[1], zeta)).v;
    } else {
        w = 0.0;
    }
}

// CGrid_Velocity.interp (_xinterpolators.py:193-332)
template <class FT, int KIND, bool TYPED>
PK_DEV void cgrid_velocity(const DGrid& g, const Coords* mc, const DField& U, const DField& V, const DField* W, const GPos& p,
                           double ypos, bool ypos_f32, double& u, double& v, double& w, bool& u32, bool& v32) {
    u32 = v32 = false;
    const int xi = p.xi, yi = p.yi, zi = p.zi;
    const double xsi = p.xsi, eta = p.eta, zeta = p.zeta;
    const int ydim = U.ny, xdim = U.nx, zdim = U.nz;
    double px[4], py[4];
    const bool rect = (KIND < 0) ? (g.kind == 0) : (KIND == 0);
    const bool cc_on = !rect && mc && mc->cc.key;
    const int cell = yi * g.nx + xi;
    if (rect) {
        const double* lon = mc ? mc->lon : g.lon;
        const double* lat = mc ? mc->lat : g.lat;
        px[0] = lon[xi]; px[1] = lon[xi + 1]; px[2] = px[1]; px[3] = px[0];
        py[0] = lat[yi]; py[1] = py[0]; py[2] = lat[yi + 1]; py[3] = py[2];
    } else if (cc_on && mc->cc.key[0] == cell) {  // the search that found (yi, xi) left its corner nodes in the lane's LDS cache
        const double* nd = mc->cc.nodes;
#pragma unroll
        for (int k = 0; k < 4; k++) { px[k] = nd[(k * 2 + 0) * CC_LANES]; py[k] = nd[(k * 2 + 1) * CC_LANES]; }
    } else {
        const double* r0 = g.node_tab + ((int64_t)yi * g.nx + xi) * 5;  // same lines the point-in-cell test just read
        const double* r1 = r0 + (int64_t)g.nx * 5;
        ldpair(r0, px[0], py[0]); ldpair(r0 + 5, px[1], py[1]);
        ldpair(r1, px[3], py[3]); ldpair(r1 + 5, px[2], py[2]);
    }
    const bool cf32x = TYPED && g.lon_f32;
    if (g.spherical) {  // :230-233
        if (cf32x) {
#pragma unroll
            for (int k = 0; k < 4; k++) px[k] = (double)(pymod360f((float)px[k] + 180.0f) - 180.0f);
#pragma unroll
            for (int k = 1; k < 4; k++)
                if ((float)px[k] - (float)px[0] > 180.0f) px[k] = (double)((float)px[k] - 360.0f);
#pragma unroll
            for (int k = 1; k < 4; k++)
                if (-(float)px[k] + (float)px[0] > 180.0f) px[k] = (double)((float)px[k] + 360.0f);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) px[k] = pymod360(px[k] + 180.0) - 180.0;
#pragma unroll
            for (int k = 1; k < 4; k++)
                if (px[k] - px[0] > 180) px[k] = px[k] - 360;
#pragma unroll
            for (int k = 1; k < 4; k++)
                if (-px[k] + px[0] > 180) px[k] = px[k] + 360;
        }
    }
    if (TYPED) {  // float32 coordinate arrays somewhere in the fieldset: carry NumPy's dtype through every operation
        cgrid_velocity_typed<FT>(g, U, V, W, p, px, py, ypos, ypos_f32, u, v, w, u32, v32);
        return;
    }
    // The staggered field values are requested FIRST and the edge-length geometry (4 cos + 4 sqrt, ~1000 cycles of fp64) is
    // computed while they are in flight: on a cache miss the wave would otherwise sit out a second global round trip after the
    // one of the cell search (the kernel waits on memory for a third of its wave cycles at 2 waves / SIMD).
    const int yi_o = clampi(yi + g.off_y, 0, ydim - 1), xi_1 = clampi(xi + 1, 0, xdim - 1);
    const int yi_1 = clampi(yi + 1, 0, ydim - 1), xi_o = clampi(xi + g.off_x, 0, xdim - 1);
    // the six staggered values of this cell at level ti (and ti+1): U at the x-faces (:272-279), V at the y-faces
    // (:281-288), W at the two z-faces (:316-328, clipped with U's z extent like the reference)
    const bool lenT = p.tau > 0;
    const int zi_0 = clampi(zi + g.off_z, 0, zdim - 1), zi_1 = clampi(zi + g.off_z + 1, 0, zdim - 1);
    FT rawf[12];  // kept in the field dtype until the geometry is done: a conversion here would wait for the loads
    FT* fv = cc_on ? (FT*)mc->cc.fvals : nullptr;
    // key[3] = 4*ti + 2*(W values cached) + (level ti+1 cached)
    const bool fhit = fv && mc->cc.key[1 * CC_LANES] == cell && mc->cc.key[2 * CC_LANES] == zi &&
                      (mc->cc.key[3 * CC_LANES] >> 2) == p.ti && (!lenT || (mc->cc.key[3 * CC_LANES] & 1)) &&
                      (!W || (mc->cc.key[3 * CC_LANES] & 2));
    if (fhit) {
#pragma unroll
        for (int k = 0; k < 6; k++) rawf[k] = fv[k * CC_LANES];
        if (lenT) {
#pragma unroll
            for (int k = 0; k < 6; k++) rawf[6 + k] = fv[(6 + k) * CC_LANES];
        }
    } else {
        int64_t off[6];
        const FT* dat[6] = {(const FT*)U.data, (const FT*)U.data, (const FT*)V.data, (const FT*)V.data,
                            W ? (const FT*)W->data : nullptr, W ? (const FT*)W->data : nullptr};
        const DField* fl[6] = {&U, &U, &V, &V, W, W};
        off[0] = cgrid_off(U, zi, yi_o, xi);
        off[1] = cgrid_off(U, zi, yi_o, xi_1);
        off[2] = cgrid_off(V, zi, yi, xi_o);
        off[3] = cgrid_off(V, zi, yi_1, xi_o);
        off[4] = W ? cgrid_off(*W, zi_0, yi_o, xi_o) : 0;
        off[5] = W ? cgrid_off(*W, zi_1, yi_o, xi_o) : 0;
#pragma unroll
        for (int k = 0; k < 6; k++) rawf[k] = (k < 4 || W) ? dat[k][slot_off(*fl[k], p.ti) + off[k]] : (FT)0;
        if (lenT) {
#pragma unroll
            for (int k = 0; k < 6; k++)
                rawf[6 + k] = (k < 4 || W) ? dat[k][slot_off(*fl[k], mini(p.ti + 1, fl[k]->nt - 1)) + off[k]] : (FT)0;
        }
    }
    constexpr bool cf32 = false;  // float64 coordinates from here on
    // einsum("ij,ji->i", phi2D_lin(eta, xsi), py): products formed for all four corners, summed in corner order.  With
    // float32 xsi/eta arrays (p.w32) the phi entries and 1 - xsi, 1 - eta, eta - 1, xsi - 1 are float32 results.
    const bool w32 = p.w32;
    const float xf = (float)xsi, ef = (float)eta;
    const double omx = w32 ? (double)(1.0f - xf) : 1 - xsi;
    const double ome = w32 ? (double)(1.0f - ef) : 1 - eta;
#define PK_PHI_DOT64(e, x_) \
    (((((1 - (x_)) * (1 - (e))) * py[0] + ((x_) * (1 - (e))) * py[1]) + ((x_) * (e)) * py[2]) + ((1 - (x_)) * (e)) * py[3])
#define PK_PHI_DOT32(e, x_)                                                                                     \
    (((((double)((1.0f - (x_)) * (1.0f - (e)))) * py[0] + ((double)((x_) * (1.0f - (e)))) * py[1]) +              \
      ((double)((x_) * (e))) * py[2]) + ((double)((1.0f - (x_)) * (e))) * py[3])
#define PK_PHI_DOT(e, x_) (w32 ? PK_PHI_DOT32((float)(e), (float)(x_)) : PK_PHI_DOT64(e, x_))
    const double c1 = geodetic_distance(g, py[0], py[1], px[0], px[1], PK_PHI_DOT(0.0, xsi), cf32);
    const double c2 = geodetic_distance(g, py[1], py[2], px[1], px[2], PK_PHI_DOT(eta, 1.0), cf32);
    const double c3 = geodetic_distance(g, py[2], py[3], px[2], px[3], PK_PHI_DOT(1.0, xsi), cf32);
    const double c4 = geodetic_distance(g, py[3], py[0], px[3], px[0], PK_PHI_DOT(eta, 0.0), cf32);
#undef PK_PHI_DOT64
#undef PK_PHI_DOT32
#undef PK_PHI_DOT
    if (!fhit) {  // fill the lane's cache slot now that the geometry has covered the latency of the loads
        if (fv) {
            mc->cc.key[1 * CC_LANES] = -1;
#pragma unroll
            for (int k = 0; k < 6; k++) fv[k * CC_LANES] = rawf[k];
            if (lenT) {
#pragma unroll
                for (int k = 0; k < 6; k++) fv[(6 + k) * CC_LANES] = rawf[6 + k];
            }
            mc->cc.key[2 * CC_LANES] = zi;
            mc->cc.key[3 * CC_LANES] = 4 * p.ti + (W ? 2 : 0) + (lenT ? 1 : 0);
            mc->cc.key[1 * CC_LANES] = cell;
        }
    }
    double raw[6];
#pragma unroll
    for (int k = 0; k < 6; k++) raw[k] = lenT ? (double)rawf[k] * (1 - p.tau) + (double)rawf[6 + k] * p.tau : (double)rawf[k];
    const double ua = raw[0], ub = raw[1], va = raw[2], vb = raw[3];
    const double U0 = ua * c4, U1 = ub * c2;
    const double Uvel = omx * U0 + xsi * U1;
    const double V0 = va * c1, V1 = vb * c3;
    const double Vvel = ome * V0 + eta * V1;
    // _compute_jacobian_determinant (utils/interpolation.py:188-198)
    const double dxs0 = w32 ? (double)(ef - 1.0f) : eta - 1, dxs1 = ome, dxs2 = eta, dxs3 = -eta;
    const double det0 = w32 ? (double)(xf - 1.0f) : xsi - 1, det1 = -xsi, det2 = xsi, det3 = omx;
    const double dxdxsi = ((dxs0 * px[0] + dxs1 * px[1]) + dxs2 * px[2]) + dxs3 * px[3];
    const double dxdeta = ((det0 * px[0] + det1 * px[1]) + det2 * px[2]) + det3 * px[3];
    const double dydxsi = ((dxs0 * py[0] + dxs1 * py[1]) + dxs2 * py[2]) + dxs3 * py[3];
    const double dydeta = ((det0 * py[0] + det1 * py[1]) + det2 * py[2]) + det3 * py[3];
    double jac = dxdxsi * dydeta - dxdeta * dydxsi;
    if (g.spherical) jac = jac * g.deg2m;
    const double A = -ome * Uvel - omx * Vvel;
    const double B = ome * Uvel - xsi * Vvel;
    const double C = eta * Uvel + xsi * Vvel;
    const double D = -eta * Uvel + omx * Vvel;
    const Recip rjac = make_recip(jac);  // two quotients per denominator: one reciprocal each (div_shared == the hardware `/`)
    double uu = div_shared(A * px[0] + B * px[1] + C * px[2] + D * px[3], rjac);
    double vv = div_shared(A * py[0] + B * py[1] + C * py[2] + D * py[3], rjac);
    if (g.spherical) {  // :311-314 (both components divided by deg2m*cos(lat))
        double conv;
        if (ypos_f32) conv = (double)((float)g.deg2m * cosf((float)ypos * DEG2RADF));
        else conv = g.deg2m * cos_lat(ypos * DEG2RAD);
        const Recip rconv = make_recip(conv);
        uu = div_shared(uu, rconv);
        vv = div_shared(vv, rconv);
    }
    u = uu;
    v = vv;
    if (W) {  // :316-328
        w = raw[4] * (1 - zeta) + raw[5] * zeta;
    } else {
        w = 0.0;
    }
}

// one value of a (possibly packed) field at integer indices, linear in time (shared by XNearest / CGrid_Tracer)
template <class FT>
PK_DEV double point_value(const DField& f, const GPos& p, int zf, int yf, int xf) {
    const FT* d = (const FT*)f.data;
    const int64_t off = ((int64_t)zf * f.st_z + (int64_t)yf * f.st_y + (int64_t)xf * f.st_x) * f.ncomp;
    double v = ldv(d, slot_off(f, p.ti) + off);
    if (p.tau > 0) v = v * (1 - p.tau) + ldv(d, slot_off(f, mini(p.ti + 1, f.nt - 1)) + off) * p.tau;
    return v;
}
// XNearest.interp (_xinterpolators.py:505-553)
template <class FT>
PK_DEV double xnearest(const DField& f, const GPos& p) {
    const int zf = p.zeta <= 0.5 ? p.zi : mini(p.zi + 1, f.nz - 1);
    const int yf = p.eta <= 0.5 ? p.yi : mini(p.yi + 1, f.ny - 1);
    const int xf = p.xsi <= 0.5 ? p.xi : mini(p.xi + 1, f.nx - 1);
    return point_value<FT>(f, p, zf, yf, xf);
}
// CGrid_Tracer.interp (_xinterpolators.py:335-383)
template <class FT>
PK_DEV double cgrid_tracer(const DGrid& g, const DField& f, const GPos& p) {
    return point_value<FT>(f, p, clampi(p.zi + g.off_z, 0, f.nz - 1), clampi(p.yi + g.off_y, 0, f.ny - 1), clampi(p.xi + g.off_x, 0, f.nx - 1));
}
// XLinearInvdistLandTracer.interp (_xinterpolators.py:556-613).  lenT / lenZ are batch-global in the reference and, unlike in
// XLinear, every gathered corner takes part: per particle here unless force_lent / force_lenz (1 / 2) give the value of the
// batch (pk_eval: one call is one batch).  `values` keeps XLinear's dtype: its assignments cast to it.
template <class FT>
PK_DEV double xlinear_invdist(const DField& f, const GPos& p, int force_lent, int force_lenz) {
    bool v32 = false;
    double value = xlinear<FT>(f, make_corners(f, p), p, &v32);
    const FT* d = (const FT*)f.data;
    const int lenT = force_lent ? force_lent : (p.tau > 0 ? 2 : 1), lenZ = force_lenz ? force_lenz : (p.zeta > 0 ? 2 : 1);
    const double zero_tol = sizeof(FT) == 4 ? (double)(float)1e-8 : 1e-8;
    double val = 0.0, w_sum = 0.0, exact_vals = 0.0;
    int nb_land = 0;
    bool has_exact = false;
    for (int it = 0; it < lenT; it++)
        for (int iz = 0; iz < lenZ; iz++)
            for (int iy = 0; iy < 2; iy++)
                for (int ix = 0; ix < 2; ix++) {
                    const int tt = it ? mini(p.ti + 1, f.nt - 1) : p.ti;
                    const int zz = iz ? mini(p.zi + 1, f.nz - 1) : p.zi;
                    const int yy = iy ? mini(p.yi + 1, f.ny - 1) : p.yi;
                    const int xx = ix ? mini(p.xi + 1, f.nx - 1) : p.xi;
                    const double c = ldv(d, slot_off(f, tt) + ((int64_t)zz * f.st_z + (int64_t)yy * f.st_y + (int64_t)xx * f.st_x) * f.ncomp);
                    const bool land = fabs(c) <= zero_tol;
                    nb_land += land;
                    const double dist2 = (p.eta - iy) * (p.eta - iy) + (p.xsi - ix) * (p.xsi - ix);
                    const double inv = 1.0 / dist2;
                    val += land ? 0.0 : c * inv;
                    w_sum += land ? 0.0 : inv;
                    if (dist2 == 0 && !land) {
                        exact_vals = sizeof(FT) == 4 ? (double)((float)exact_vals + (float)c) : exact_vals + c;
                        has_exact = true;
                    }
                }
    if (nb_land == 4 * lenZ * lenT) return 0.0;
    if (nb_land > 0) {
        value = val / w_sum;
        if (has_exact) value = exact_vals;
        if (v32) value = (double)(float)value;
    }
    return value;
}

// NaN -> ErrorInterpolation (field.py:373-378) then out-of-bounds -> 0 (field.py:359-370)
// The reference interpolates wrapped-around garbage for out-of-bounds lanes before zeroing it: finite, unless a barycentric
// coordinate is non-finite (position +-inf / NaN) -> NaN -> ErrorInterpolation.  `oob_flags`: bit 0 out of bounds, bit 1 that
// non-finite case; computed once right after the search so that the coordinates need not stay live.
PK_DEV int oob_flags(const GPos& p) {
    const bool oob = p.xi < 0 || p.yi < 0 || p.zi < 0;
    const bool bad = oob && !(isfinite(p.xsi) && isfinite(p.eta) && isfinite(p.zeta) && isfinite(p.tau));
    return (oob ? 1 : 0) | (bad ? 2 : 0);
}
PK_DEV double finish_value(PCtx& c, int flags, double v) {
    if (flags & 2) v = NAN;
    if (v != v && c.state < PK_ERRORINTERPOLATION) c.state = PK_ERRORINTERPOLATION;
    if (flags & 1) {
        v = 0.0;
        c.oob = true;
    }
    return v;
}
PK_DEV double finish_value(PCtx& c, const GPos& p, double v) { return finish_value(c, oob_flags(p), v); }

// VectorField.eval (field.py:250-304). INTERP: 0 XLinear_Velocity, 1 CGrid_Velocity. pos_f32: z,y,x come
// straight from float32 particle storage (NumPy then evaluates cos(lat) in float32).
template <class FT, int KIND, int INTERP, bool TYPED>
PK_DEV void eval_uvw(const KArgs& a, const Coords& mc, PCtx& c, bool want_w, double t, double z, double y,
                     double x, bool pos_f32, double& u, double& v, double& w, SearchMemo* memo = nullptr) {
    const DField& U = kfield(a, a.prm.fU);
    const DField& V = kfield(a, a.prm.fV);
    const DGrid& g = kgrid(a, U.grid);
    GPos p;
    u = v = w = 0.0;
    c.u32 = c.v32 = false;
    const int klo = c.klo++;
    if (U.has_time_interval) {
        const int li = twe_listed_index(a, c.it, klo);
        if (li >= 0) {  // somebody leaves the time interval at this sample: the whole view takes the code (field.py:31-44)
            twe_justify(a, li, t, U.tlen);
            c.state = PK_ERROROUTSIDETIMEINTERVAL;
            return;
        }
    }
    if (!time_search(U, mc.time, t, c.ht, p)) {  // field.py:303-304 -> _deal_with_errors: state := 70, zeros
        c.state = PK_ERROROUTSIDETIMEINTERVAL;
        twe_note_all(a, c.it, klo);
        return;
    }
    c.ht = p.ti;
    const bool use_guess = take_first_eval(c, U.grid) ? (a.prm.have_guess0 != 0) : true;
    int32_t ei = ei_get(c, U.grid);
    grid_search<KIND, TYPED>(g, &mc, z, y, x, pos_f32, &ei, c, use_guess, p);
    ei_set(c, U.grid, ei);
    if (memo) {
        memo->p = p;
        memo->grid = (p.xi < 0 || p.yi < 0 || p.zi < 0) ? -1 : U.grid;
    }
    const int flags = oob_flags(p);
    const bool oob = flags & 1;
    double uu = 0, vv = 0, ww = 0;
    if (!oob) {
        const DField* W = (want_w && a.prm.fW >= 0) ? &kfield(a, a.prm.fW) : nullptr;
        if (INTERP == 1) {
            cgrid_velocity<FT, KIND, TYPED>(g, &mc, U, V, W, p, y, pos_f32, uu, vv, ww, c.u32, c.v32);
        } else if (INTERP == 2) {  // XFreeslip (interp_uv == 2) / XPartialslip (3)
            const bool freeslip = a.prm.interp_uv == 2;
            slip_velocity<FT>(g, U, V, W, p, y, pos_f32, freeslip ? 1.0 : 0.5, freeslip ? 0.0 : 0.5, a.prm.force_lenz, uu, vv, ww, c.u32, c.v32);
        } else {  // XLinear_Velocity.interp (_xinterpolators.py:169-190)
            const Corners k = make_corners(U, p);
            bool u32 = false, v32 = false;
            uu = xlinear<FT>(U, k, p, &u32);
            PK_FIELD_FENCE();
            vv = same_layout(U, V) ? xlinear<FT>(V, k, p, &v32) : xlinear<FT>(V, make_corners(V, p), p, &v32);
            PK_FIELD_FENCE();
            if (W) ww = same_layout(U, *W) ? xlinear<FT>(*W, k, p) : xlinear<FT>(*W, p);
            if (g.spherical) {  // in-place /= : a float32 u / v array is divided and stored in float32
                double conv;
                if (pos_f32) conv = (double)((float)g.deg2m * cosf((float)y * DEG2RADF));
                else conv = g.deg2m * cos_lat(y * DEG2RAD);
                if (u32) uu = pos_f32 ? (double)((float)uu / (float)conv) : (double)(float)(uu / conv);
                else uu /= conv;
                if (v32) vv = (double)((float)vv / (float)g.deg2m);
                else vv /= g.deg2m;
            }
            c.u32 = u32;
            c.v32 = v32;
        }
    }
    u = finish_value(c, flags, uu);
    v = finish_value(c, flags, vv);
    w = finish_value(c, flags, ww);
}

// Field.eval for a scalar field (field.py:145-195): XLinear or XConstantField
template <class FT, bool TYPED>
PK_DEV double eval_scalar(const KArgs& a, const Coords& mc, PCtx& c, int fidx, double t, double z, double y,
                          double x, bool pos_f32, const SearchMemo* memo = nullptr) {
    const DField& f = kfield(a, fidx);
    const DGrid& g = kgrid(a, f.grid);
    const bool on_main = (f.grid == a.main_grid);
    GPos p;
    const double* time = (on_main && f.time == kfield(a, a.main_field).time) ? mc.time : f.time;
    const int klo = c.klo++;
    if (f.has_time_interval) {
        const int li = twe_listed_index(a, c.it, klo);
        if (li >= 0) {  // (see eval_uvw)
            twe_justify(a, li, t, f.tlen);
            c.state = PK_ERROROUTSIDETIMEINTERVAL;
            return 0.0;
        }
    }
    if (!time_search(f, time, t, on_main ? c.ht : 0, p)) {
        c.state = PK_ERROROUTSIDETIMEINTERVAL;
        twe_note_all(a, c.it, klo);
        return 0.0;
    }
    const bool use_guess = take_first_eval(c, f.grid) ? (a.prm.have_guess0 != 0) : true;
    int32_t ei = ei_get(c, f.grid);
    // (t, z, y, x) is the point of the kernel's velocity sample: see SearchMemo for the two conditions
    const bool curv = g.kind == 1;
    if (memo && memo->grid == f.grid && (!curv || (on_main && c.hyx_valid && c.hy == memo->p.yi && c.hx == memo->p.xi))) {
        const int ti = p.ti;
        const double tau = p.tau;
        p = memo->p;
        p.ti = ti;
        p.tau = tau;
        if (curv) {  // a guessed search that hits: float64 coordinates, no float32 arrays
            p.xsi = p.xsi_raw;
            p.eta = p.eta_raw;
            p.w32 = p.x32 = p.e32 = false;
        }
        grid_search_finish(g, on_main, curv, &ei, c, p);
    } else {
        grid_search<-1, TYPED>(g, on_main ? &mc : nullptr, z, y, x, pos_f32, &ei, c, use_guess, p);
    }
    ei_set(c, f.grid, ei);
    double v = 0.0;
    if (!(p.xi < 0 || p.yi < 0 || p.zi < 0)) {
        const bool f64 = f.dtype == PK_F64;
        switch (f.is_const) {
            case 1: v = f64 ? ((const double*)f.data)[f.comp] : (double)((const float*)f.data)[f.comp]; break;
            case 2: v = f64 ? xnearest<double>(f, p) : xnearest<float>(f, p); break;
            case 3: v = f64 ? cgrid_tracer<double>(g, f, p) : cgrid_tracer<float>(g, f, p); break;
            case 4: v = f64 ? xlinear_invdist<double>(f, p, a.prm.force_lent, a.prm.force_lenz) : xlinear_invdist<float>(f, p, a.prm.force_lent, a.prm.force_lenz); break;
            default: v = f64 ? xlinear<double>(f, p) : xlinear<float>(f, p); break;
        }
    }
    return finish_value(c, p, v);
}

// ---- counter-based RNG for the stochastic kernels --------------------------------------------------------
// The reference draws from NumPy's global MT19937 stream (_advectiondiffusion.py:37-38), which no parallel
// engine can reproduce.  Here: Philox4x32-10, key = (seed, kernel slot), counter = (particle_id, bits of the
// particle's time) -> two standard normals by Box-Muller.  Stateless, so trajectories do not depend on
// launch partitioning or on how particles are sharded over GPUs.
PK_DEV void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
PK_DEV void normal_pair(uint64_t seed, int kslot, int64_t particle_id, double t, double& z0, double& z1) {
    uint32_t k[2] = {(uint32_t)seed ^ (0x9E3779B9u * (uint32_t)(kslot + 1)), (uint32_t)(seed >> 32)};
    const uint64_t tb = (uint64_t)__double_as_longlong(t);
    uint32_t c[4] = {(uint32_t)particle_id, (uint32_t)((uint64_t)particle_id >> 32), (uint32_t)tb, (uint32_t)(tb >> 32)};
#pragma unroll
    for (int r = 0; r < 10; r++) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    const uint64_t a = ((uint64_t)c[1] << 32) | c[0], b = ((uint64_t)c[3] << 32) | c[2];
    const double u1 = ((double)(a >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)(b >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u1));
    const double th = 6.283185307179586476925286766559 * u2;
    double s, co;
    sincos_geo(th, s, co);  // (th in [0, 2 pi): < 1 ulp like the library's, without its large-argument path)
    z0 = r * co;
    z1 = r * s;
}

}  // namespace pk
