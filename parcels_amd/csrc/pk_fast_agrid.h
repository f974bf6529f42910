// pk_fast_agrid.h -- VectorField.eval for the headline configuration: XLinear_Velocity on a rectilinear A-grid with
// float64 coordinates (BASELINE config 2; field.py:250-304, index_search.py:20-91, _xinterpolators.py:25-190).
//
// Same arithmetic, expression by expression, as the general eval_uvw<FT, 0, 0, false> of pk_device.h -- the parity tests run
// both against the same fixtures -- but laid out for the CDNA4 issue ports:
//   * everything the general path re-derives per evaluation from the field / grid descriptors (strides, extents, slot
//     arithmetic, layout comparisons) is folded into one small `FastA` block of wave-uniform values by the host;
//   * the 1-D coordinate vectors are staged into LDS as {a[i], 1 / (a[i+1] - a[i])} pairs: the barycentric quotient
//     (x - a[i]) / (a[i+1] - a[i]) is formed with the correctly rounded reciprocal and two fused residual steps (Markstein's
//     theorem: identical to the IEEE quotient) -- 5 fp64 issue slots instead of the ~14 of v_div_scale / v_rcp / v_div_fmas;
//   * the time level, and whether the second time / depth level takes part, is made wave-uniform with a readfirstlane
//     waterfall (one pass unless particles of one wavefront sit on different time levels): level base addresses live in
//     SGPRs, every corner-pair load is `global_load_dwordx4 v, v_off, s[base]` with ONE shared 32-bit lane offset, and the
//     lenT / lenZ selects of the general path become scalar branches;
//   * the particle storage dtype is a template parameter.
#pragma once
#include "pk_device.h"

namespace pk {

// interleaved coordinate table entry: node value and reciprocal of the width of the cell that starts there
typedef double pk_tab2 __attribute__((ext_vector_type(2)));

struct FastTabs {  // LDS-resident {a, 1/width} tables of the main grid and the velocity field's time axis
    const pk_tab2* time;
    const pk_tab2* depth;
    const pk_tab2* lat;
    const pk_tab2* lon;
    pk_tab2* blk;  // this lane's corner-block cache (FCtx::bei): 4 x 16 bytes at stride FAST_WG, NULL = none
    uint32_t fl;   // FA_* bits of the wave-uniform yes / no questions of one evaluation (fast_flags)
    // Scalars of FastA that every evaluation reads, PINNED in scalar registers by the kernel (pin_scalars: an opaque asm makes them values
    // the allocator may spill to a VGPR lane -- one v_readlane where it is used -- but can no longer re-load from the kernel-argument
    // segment: round 5's kernel did `s_load_dwordx2 ... 0x328` (tlen) + `s_waitcnt lgkmcnt(0)` in every evaluation, a scalar-cache
    // round trip of ~200 cycles that also drains the LDS counter)
    double tlen;
    int32_t gny, gnx;
};
PK_DEV void pin_scalars(FastTabs& T, const FastA& F) {
    T.tlen = F.tlen;
    T.gny = F.gny;
    T.gnx = F.gnx;
#ifdef PK_FAST_PIN  // measured (profiles/r06c_*): the pinned build is no faster than the one that re-loads them (7.92-7.97 vs 7.60-7.63 ms on
                    // two boxes whose round-5 baselines differ by 2 %: within noise or slightly worse) -- off
    asm volatile("" : "+s"(T.tlen), "+s"(T.gny), "+s"(T.gnx));
#endif
}
// The wave-uniform booleans of an evaluation as bits of ONE scalar register.  Kept as separate loop-invariant i1 values the compiler
// holds each of them as a 64-bit lane mask (`s_cselect_b64 -1, 0`): two SGPRs per question, 46 SGPRs spilled to VGPR lanes in round 5's
// kernel and a `v_readlane_b32` -- a VALU instruction -- for every use inside the stage loop.  eval_uvw_fast re-reads the word through
// an opaque scalar at its top, so every test is an `s_bitcmp` + `s_cbranch_scc` where it is used.
enum : uint32_t {
    FA_TI = 1u, FA_Z = 2u, FA_Y = 4u, FA_X = 8u, FA_SPH = 16u, FA_RING = 32u, FA_BLK = 64u,
    FA_NT2 = 128u, FA_NZ2 = 256u, FA_NY2 = 512u, FA_NX2 = 1024u,  // the axis has at least two nodes (index_search.py:45-46)
    FA_WIN = 2048u, FA_FWD = 4096u, FA_MAXIT = 8192u               // step loop: some ring streams; dt0 > 0; an iteration limit is set
};
PK_DEV uint32_t fast_flags(const FastA& F) {
    return (F.has_ti ? FA_TI : 0u) | (F.has_z ? FA_Z : 0u) | (F.has_y ? FA_Y : 0u) | (F.has_x ? FA_X : 0u) | (F.spherical ? FA_SPH : 0u) |
           (F.nslots < F.nt ? FA_RING : 0u) | (F.lds_blk != 0 ? FA_BLK : 0u) | (F.nt >= 2 ? FA_NT2 : 0u) | (F.gnz >= 2 ? FA_NZ2 : 0u) |
           (F.gny >= 2 ? FA_NY2 : 0u) | (F.gnx >= 2 ? FA_NX2 : 0u);
}
// Square root, quotient and reciprocal WITHOUT the library's range scaling and final fix-up (PK_CG_LEAN / PK_FAST_LEAN, round 6; the tolerance
// argument of pk_fast_cgrid.h: sincos_near): the operands here are lengths in metres, squared or not, determinants and Jacobians of cells -- far from the
// subnormal and overflow ranges the library's 20-instruction sqrt and 10-instruction division guard -- and one Goldschmidt / Newton
// step + one residual correction on the hardware's v_rsq_f64 / v_rcp_f64 leaves < 1 ulp (not always the correctly rounded bit).
#ifndef PK_CG_LEAN
#define PK_CG_LEAN 1
#endif
PK_DEV double sqrt_lean(double x) {  // x > 0 (0, negative, NaN: the caller selects)
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    return fma(fma(-g, g, x), h, g);
}
PK_DEV double rcp_lean(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(r, fma(-b, r, 1.0), r);
    return fma(r, fma(-b, r, 1.0), r);
}
PK_DEV double div_lean(double a, double b) {
    const double r = rcp_lean(b), q = a * r;
    return fma(fma(-b, q, a), r, q);
}

// Barycentric coordinate of x in the cell [a, a1] of an {a, 1 / width} table.  PK_FAST_LEAN (round 6, default): (x - a) * RN(1 / width), one
// multiplication, within 1.5 ulp of the reference's (x - a) / (a1 - a) and never above 1 for x <= a1 (n * RN(1 / d) <= RN(1 + 2^-53) = 1);
// 0: the correctly rounded quotient (div_by_recip: five fused operations + a guard), the bits of the general program.  The index, the
// out-of-bounds codes and "exactly on a node / level" (bc == 0) do not depend on the choice.
#ifndef PK_FAST_LEAN
#define PK_FAST_LEAN 1
#endif
// (Contracting the lerps and bilinear sums of this kernel into fused multiply-adds -- `#pragma clang fp contract(fast)` in lerp_rows / uvw_fast,
// 70 of its 286 fp64 instructions -- was measured and is not done: 6.99 -> 6.98 ms, profiles/r06v_contraction_ab.txt.  The C-grid
// evaluation is contracted: pk_fast_cgrid.h, PK_CG_FMA.)
PK_DEV double fast_bary(double x, double a, double a1, double rw) {
    if constexpr (PK_FAST_LEAN != 0) return (x - a) * rw;
    else return div_by_recip(x - a, a1 - a, rw);
}
constexpr int FAST_WG = 256;                          // lanes per workgroup of advect_fast_kernel
constexpr int FAST_BLK_BYTES = FAST_WG * 8 * 8;       // LDS of the corner-block cache per workgroup (8 doubles per lane)

// clip(searchsorted(arr, x, "left") - 1, 0, n-2) over the interleaved table.  Cold (a lane that moved two cells or more, or holds a
// NaN): one bisection loop -- the unrolled three-step walk of rounds 1-5 found the same index and cost the hot path seven nesting
// levels of saved exec masks (14 SGPRs) at each of its four inlined copies.
PK_DEV int cell_index_tab(const pk_tab2* tab, int n, double x, int i) {
    (void)i;
    asm volatile("" : "+s"(n));  // (wave-uniform; opaque so that `0 < n` of the loop entry is not hoisted into a saved lane mask)
    if (x != x) return n - 2;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (tab[mid].x < x) lo = mid + 1; else hi = mid;
    }
    return clampi(lo - 1, 0, n - 2);
}

// _search_1d_array (index_search.py:20-62) for a float64 coordinate: the hinted cell is right for all but the lanes that just
// crossed a cell edge.  `cell` (in / out): a valid cell index 0..n-2, the hint on entry and the cell found on exit; idx: that
// cell or the out-of-bounds code.  The cell test is NaN-proof as written: a NaN fails it unless the hint already is the
// last cell (NumPy sorts NaN last), and cell_index_tab answers n-2.
// FLAGGED: the caller answers "n >= 2" itself (`two`, a bit of FastTabs::fl tested where it is used).
template <bool FLAGGED = false>
PK_DEV void fast_search(const pk_tab2* tab, int n, double first, double last, double x, int& cell, int& idx, double& bc, bool two = true) {
    if (FLAGGED ? !two : n < 2) {  // :45-46
        idx = 0;
        bc = 0.0;
        return;
    }
    int i = cell;
    pk_tab2 e = tab[i];
    double a1 = tab[i + 1].x;
    const bool lo_ok = e.x < x || i == 0, hi_ok = x <= a1 || i == n - 2;
    if (!(lo_ok && hi_ok)) {
        // Some lane of nearly every wavefront crosses a cell edge in nearly every evaluation, so this block is hot: the neighbour
        // cell the failed test points at is probed without a loop (one LDS round trip for the whole wave); only lanes that moved
        // further, or hold a NaN, take the walk + bisect of cell_index_tab.  (!hi_ok implies i < n-2, !lo_ok implies i > 0.)
        int j = i + (hi_ok ? 0 : 1) - (lo_ok ? 0 : 1);
        pk_tab2 e2 = tab[j];
        double b1 = tab[j + 1].x;
        asm volatile("" : "+v"(b1));  // both reads in flight together (not the second one behind the first compare's short circuit)
        if (__builtin_expect(!((e2.x < x || j == 0) && (x <= b1 || j == n - 2)), 0)) {
            j = cell_index_tab(tab, n, x, j);
            e2 = tab[j];
            b1 = tab[j + 1].x;
        }
        i = j;
        e = e2;
        a1 = b1;
    }
    bc = fast_bary(x, e.x, a1, e.y);
    cell = i;
    // :55-58.  A point left of the first node is in cell 0, one right of the last node in cell n-2 (the clip of :47), so only lanes in
    // an edge cell pay the two fp64 compares (a divergent block most wavefronts skip: the particles live in the interior)
    if (__builtin_expect(i == 0 || i == n - 2, 0)) {
        if (x < first) i = LEFT_OUT_OF_BOUNDS;
        if (x > last) i = RIGHT_OUT_OF_BOUNDS;
    }
    idx = i;
}

// Two axes at once (lat and lon of one evaluation; both with n >= 2): the same answers as two fast_search calls, but the LDS round trips of
// the two axes overlap -- all four hint-cell reads are in flight together, and so are the four reads of the neighbour probes when some lane
// crossed an edge on either axis (which happens in nearly every wave-evaluation: round 5's kernel walked through up to six dependent
// ~100-cycle LDS round trips per evaluation here; the PMC pass showed 0.36 of its wave cycles in s_waitcnt and no gain from fewer VALU
// instructions alone, profiles/r06a_c2_flags_pmc.md).
PK_DEV void fast_search2(const pk_tab2* taba, int na, double firsta, double lasta, double xa, int& cella, int& idxa, double& bca,
                         const pk_tab2* tabb, int nb, double firstb, double lastb, double xb, int& cellb, int& idxb, double& bcb) {
    int ia = cella, ib = cellb;
    pk_tab2 ea = taba[ia], eb = tabb[ib];
    double a1a = taba[ia + 1].x, a1b = tabb[ib + 1].x;
    const bool lo_a = ea.x < xa || ia == 0, hi_a = xa <= a1a || ia == na - 2;
    const bool lo_b = eb.x < xb || ib == 0, hi_b = xb <= a1b || ib == nb - 2;
    if (!(lo_a && hi_a && lo_b && hi_b)) {
        // (an axis that passed keeps its cell: j == i, and the probe below re-reads the same entries)
        int ja = ia + (hi_a ? 0 : 1) - (lo_a ? 0 : 1), jb = ib + (hi_b ? 0 : 1) - (lo_b ? 0 : 1);
        pk_tab2 e2a = taba[ja], e2b = tabb[jb];
        double b1a = taba[ja + 1].x, b1b = tabb[jb + 1].x;
        asm volatile("" : "+v"(b1a), "+v"(b1b));  // all four reads issued before the first compare (not sunk behind its short circuit)
        if (__builtin_expect(!((e2a.x < xa || ja == 0) && (xa <= b1a || ja == na - 2)), 0)) {
            ja = cell_index_tab(taba, na, xa, ja);
            e2a = taba[ja];
            b1a = taba[ja + 1].x;
        }
        if (__builtin_expect(!((e2b.x < xb || jb == 0) && (xb <= b1b || jb == nb - 2)), 0)) {
            jb = cell_index_tab(tabb, nb, xb, jb);
            e2b = tabb[jb];
            b1b = tabb[jb + 1].x;
        }
        ia = ja; ea = e2a; a1a = b1a;
        ib = jb; eb = e2b; a1b = b1b;
    }
    bca = fast_bary(xa, ea.x, a1a, ea.y);
    bcb = fast_bary(xb, eb.x, a1b, eb.y);
    cella = ia;
    cellb = ib;
    if (__builtin_expect(ia == 0 || ia == na - 2 || ib == 0 || ib == nb - 2, 0)) {  // :55-58, see fast_search
        if (xa < firsta) ia = LEFT_OUT_OF_BOUNDS;
        if (xa > lasta) ia = RIGHT_OUT_OF_BOUNDS;
        if (xb < firstb) ib = LEFT_OUT_OF_BOUNDS;
        if (xb > lastb) ib = RIGHT_OUT_OF_BOUNDS;
    }
    idxa = ia;
    idxb = ib;
}

// The two x-corners of one (level, z, y) row at byte offset `off` from a wave-uniform base: one wide load in saddr form.
// PK_ABLATE_FIELDS (measurement builds only, results are wrong on purpose; tools/build_variant.sh): what an ideal LDS field tile
//  could buy at most.  1 = every lane reads the SAME global address (the 16 loads become one cache line per wavefront: no L2
// / Infinity-Cache latency spread, no TA divergence); 2 = the corner values come out of LDS (ds_read_b128 from the staged coordinate
// tables: a tile that costs nothing to maintain).
#ifndef PK_ABLATE_FIELDS
#define PK_ABLATE_FIELDS 0
#endif
template <class FT>
PK_DEV void ldrow(const char* base, uint32_t off, double& a, double& b) {
#if PK_ABLATE_FIELDS == 1
    ldpair(reinterpret_cast<const FT*>(base + (off & 8u)), a, b);
#elif PK_ABLATE_FIELDS == 2
    extern __shared__ __attribute__((aligned(16))) double pk_ablate_smem[];
    const pk_tab2 v = reinterpret_cast<const pk_tab2*>(pk_ablate_smem)[(off >> 4) & 63u];
    a = v.x * 1e-9 + (double)((uintptr_t)base & 1);
    b = v.y * 1e-9;
#else
    ldpair(reinterpret_cast<const FT*>(base + off), a, b);
#endif
}

// XLinear.interp (_xinterpolators.py:112-153) for one field; LT / LZ: the second time / depth level takes part (wave-uniform,
// compile-time here so that the whole gather + interpolation of a field is ONE basic block of up to 8 wide loads).
// l0 / l1: wave-uniform bases of the two time levels; b00: byte offset of corner (zi, yi, xi) inside a level; dyb / dzb: byte
// strides to the yi+1 row / zi+1 plane (0 on an axis the field does not have).
struct Rows {  // the (up to) 16 corner values of one field: [time level][z][y] rows of two x-neighbours
    double c00, c01, c10, c11, t00, t01, t10, t11, d00, d01, d10, d11, e00, e01, e10, e11;
};
template <class FT, bool LT, bool LZ>
PK_DEV void load_rows(Rows& r, const char* l0, const char* l1, uint32_t b00, uint32_t dyb, uint32_t dzb) {
    ldrow<FT>(l0, b00, r.c00, r.c01);
    ldrow<FT>(l0 + dyb, b00, r.c10, r.c11);
    if (LT) {
        ldrow<FT>(l1, b00, r.t00, r.t01);
        ldrow<FT>(l1 + dyb, b00, r.t10, r.t11);
    }
    if (LZ) {
        ldrow<FT>(l0 + dzb, b00, r.d00, r.d01);
        ldrow<FT>(l0 + dzb + dyb, b00, r.d10, r.d11);
        if (LT) {
            ldrow<FT>(l1 + dzb, b00, r.e00, r.e01);
            ldrow<FT>(l1 + dzb + dyb, b00, r.e10, r.e11);
        }
    }
}
// lerp in t, then z, then bilinear in (eta, xsi) with the weights (1-xsi)(1-eta), xsi(1-eta), (1-xsi)eta, xsi*eta shared by U, V, W
// (lerp_rows: the t / z part -- a function of the cell and of (t, z) only; interp_rows: all of it)
template <bool LT, bool LZ>
PK_DEV void lerp_rows(const Rows& r, double tau, double omt, double zeta, double omz, double& c00, double& c01, double& c10, double& c11) {
    c00 = r.c00; c01 = r.c01; c10 = r.c10; c11 = r.c11;
    if (LT) {
        c00 = c00 * omt + r.t00 * tau; c01 = c01 * omt + r.t01 * tau;
        c10 = c10 * omt + r.t10 * tau; c11 = c11 * omt + r.t11 * tau;
    }
    if (LZ) {
        double d00 = r.d00, d01 = r.d01, d10 = r.d10, d11 = r.d11;
        if (LT) {
            d00 = d00 * omt + r.e00 * tau; d01 = d01 * omt + r.e01 * tau;
            d10 = d10 * omt + r.e10 * tau; d11 = d11 * omt + r.e11 * tau;
        }
        c00 = c00 * omz + d00 * zeta; c01 = c01 * omz + d01 * zeta;
        c10 = c10 * omz + d10 * zeta; c11 = c11 * omz + d11 * zeta;
    }
}
template <bool LT, bool LZ>
PK_DEV double interp_rows(const Rows& r, double tau, double omt, double zeta, double omz, double w00, double w01, double w10, double w11) {
    double c00, c01, c10, c11;
    lerp_rows<LT, LZ>(r, tau, omt, zeta, omz, c00, c01, c10, c11);
    return w00 * c00 + w01 * c01 + w10 * c10 + w11 * c11;
}

// U, V (and W) of one evaluation.  PK_FAST_BATCH: 8 = the corner rows of one field are requested back to back (one memory round
// trip per field), 16 = those of U and V together (more registers), 4 = left to the scheduler.  The fences keep the machine
// scheduler from sinking the loads back between the arithmetic.
#ifndef PK_FAST_BATCH
#define PK_FAST_BATCH 8
#endif
#ifndef PK_FAST_BLOCK_CACHE
#define PK_FAST_BLOCK_CACHE 1  // FCtx::bei (0: A/B builds without the corner-block cache)
#endif
template <class FT, bool D3, bool LT, bool LZ>
PK_DEV void uvw_fast(const FastA& F, int64_t o0, int64_t o1, uint32_t b00, double tau, double omt, double zeta, double omz, double w00,
                     double w01, double w10, double w11, double& uu, double& vv, double& ww, pk_tab2* blk) {
    // keep the 32-bit lane offset opaque up to here: its zero-extension must sit in the basic block of the loads for the
    // instruction selector to fold it into the `saddr + voffset` addressing mode (otherwise one 64-bit VALU add per load)
    uint32_t bo = b00;
    asm volatile("" : "+v"(bo));
    Rows ru, rv, rw;
    load_rows<FT, LT, LZ>(ru, F.U + o0, F.U + o1, bo, F.dyb, F.dzb);
#if PK_FAST_BATCH == 16
    load_rows<FT, LT, LZ>(rv, F.V + o0, F.V + o1, bo, F.dyb, F.dzb);
#endif
#if PK_FAST_BATCH >= 8
    PK_FIELD_FENCE();
#endif
    // the t / z-lerped corner blocks of U and V go to the lane's block cache (FCtx::bei) as soon as they exist; blk: wave-uniform NULL or not
    double c00, c01, c10, c11;
    lerp_rows<LT, LZ>(ru, tau, omt, zeta, omz, c00, c01, c10, c11);
    if (!D3 && blk) {
        pk_tab2 b;
        b.x = c00; b.y = c01; blk[0] = b;
        b.x = c10; b.y = c11; blk[FAST_WG] = b;
    }
    uu = w00 * c00 + w01 * c01 + w10 * c10 + w11 * c11;
#if PK_FAST_BATCH != 16
    load_rows<FT, LT, LZ>(rv, F.V + o0, F.V + o1, bo, F.dyb, F.dzb);
#if PK_FAST_BATCH >= 8
    PK_FIELD_FENCE();
#endif
#endif
    lerp_rows<LT, LZ>(rv, tau, omt, zeta, omz, c00, c01, c10, c11);
    if (!D3 && blk) {
        pk_tab2 b;
        b.x = c00; b.y = c01; blk[2 * FAST_WG] = b;
        b.x = c10; b.y = c11; blk[3 * FAST_WG] = b;
    }
    vv = w00 * c00 + w01 * c01 + w10 * c10 + w11 * c11;
    if (D3) {
        load_rows<FT, LT, LZ>(rw, F.W + o0, F.W + o1, bo, F.dyb, F.dzb);
#if PK_FAST_BATCH >= 8
        PK_FIELD_FENCE();
#endif
        ww = interp_rows<LT, LZ>(rw, tau, omt, zeta, omz, w00, w01, w10, w11);
    }
}

PK_DEV int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Per-particle evaluation context of the fast kernels.  Besides the status code and the `ei` entry of the velocity grid it holds
// the cells of the previous evaluation (always valid cell indices: the search hints) and a memo of the last time and depth
// searched: the search is a pure function of the coordinate, and the Runge-Kutta stages revisit t (stages 2 and 3 share
// t + dt/2; stage 4 of one step and stage 1 of the next share t + dt) and, in the 2-D kernels, never move z.
struct FCtx {
    int state;
    int32_t ei;
    int ht, hz, hy, hx;
    int zi;                      // memo: index (or out-of-bounds code) of the last depth searched
    double mt, mtau, mz, mzeta;  // memo keys (bitwise-equal coordinate => same answer) and barycentric values
    // Corner-block cache (2-D kernels): the t / z-lerped corner values c00..c11 of U and V -- 8 doubles in the lane's LDS slot FastTabs::blk
    // -- are a function of the cell and of (t, z) only.  Stages 2 and 3 of a Runge-Kutta step share t, stage 4 and stage 1 of the next
    // step too, their sample points lie a fraction of a cell apart and z does not move: when EVERY lane of the wavefront samples the
    // cell (`bei`, its ravelled index) at the (mt, mz) its cached block was formed at, the 16 corner loads and the level lerps are skipped
    // and the bilinear sum runs on the same values -- the same operations, so the same bits (measured: 0.998 / 1.000 of the wave-evaluations
    // of those two stage pairs on BASELINE config 2).  Invariant: bei >= 0 => the slot holds the block of cell bei at (mt, mz).
    int32_t bei;
};
constexpr int32_t FAST_NO_BLOCK = -0x7fffffff - 1;
PK_DEV void fctx_init(FCtx& c, int state, int32_t ei) {
    c.state = state;
    c.ei = ei;
    c.ht = c.hz = c.hy = c.hx = 0;
    c.zi = 0;
    c.mt = c.mz = __builtin_nan("");  // equal to nothing
    c.mtau = c.mzeta = 0.0;
    c.bei = FAST_NO_BLOCK;
}

// VectorField.eval (field.py:250-304) + XLinear_Velocity.interp (_xinterpolators.py:169-190).  PF: the sample point may come
// straight from float32 particle storage (pos_f32); D3: sample W as well.  bmode (wave-uniform; FCtx::bei): bit 0 = this sample may
// share (t, z) with the previous one -- test the cached corner block; bit 1 = the next one may share them with this one -- keep the block.
template <class FT, bool PF, bool D3>
PK_DEV void eval_uvw_fast(const KArgs& a, const FastTabs& T, FCtx& c, double t, double z, double y, double x, bool pos_f32, double& u,
                          double& v, double& w, unsigned it, int klo, int bmode = 0) {
    const FastA& F = a.fast;
    uint32_t fl = T.fl;
    asm volatile("" : "+s"(fl));  // opaque: every FA_* test below is a scalar bit test HERE, not a loop-invariant lane mask (FastTabs::fl)
    u = v = w = 0.0;
    int ti = 0;
    double tau = 0.0;
    if (fl & FA_TI) {  // _search_time_index (index_search.py:65-91); (it, klo): the key of this sample (pk_device.h: twe_note -- a launch with LISTED
                     // samples runs the general program, pk_api.hip)
        if (__builtin_expect(!(0 <= t) || !(t <= T.tlen), 0)) {
            c.state = PK_ERROROUTSIDETIMEINTERVAL;
            twe_note(a, it, klo);
            return;
        }
        if (t != c.mt) {
            int idx;
            fast_search<true>(T.time, F.nt, F.t0, F.t1, t, c.ht, idx, c.mtau, (fl & FA_NT2) != 0);  // level times start at 0 (host check): idx == c.ht
            c.mt = t;
            c.bei = FAST_NO_BLOCK;
        }
        ti = c.ht;
        tau = c.mtau;
    }
    int zi = 0, yi = 0, xi = 0;
    double zeta = 0.0, eta = 0.0, xsi = 0.0;
    if (fl & FA_Z) {
        if (!(z == c.mz)) {
            fast_search<true>(T.depth, F.gnz, F.z0, F.z1, z, c.hz, c.zi, c.mzeta, (fl & FA_NZ2) != 0);
            c.mz = z;
            c.bei = FAST_NO_BLOCK;
        }
        zi = c.zi;
        zeta = c.mzeta;
    }
#ifndef PK_FAST_SEARCH2
#define PK_FAST_SEARCH2 1
#endif
    constexpr uint32_t YX2 = FA_Y | FA_X | FA_NY2 | FA_NX2;
    if (PK_FAST_SEARCH2 && (fl & YX2) == YX2) {
        fast_search2(T.lat, T.gny, F.y0, F.y1, y, c.hy, yi, eta, T.lon, T.gnx, F.x0, F.x1, x, c.hx, xi, xsi);
    } else {
        if (fl & FA_Y) fast_search<true>(T.lat, T.gny, F.y0, F.y1, y, c.hy, yi, eta, (fl & FA_NY2) != 0);
        if (fl & FA_X) fast_search<true>(T.lon, T.gnx, F.x0, F.x1, x, c.hx, xi, xsi, (fl & FA_NX2) != 0);
    }
    // ravel_index (basegrid.py:83-152): the low 32 bits of the int64 sum are the wrapped 32-bit sum
    c.ei = (int32_t)((uint32_t)xi * F.ex + (uint32_t)yi * F.ey + (uint32_t)zi * F.ez);
    if (__builtin_expect((xi | yi | zi) < 0, 0)) {  // some index carries an out-of-bounds code (-1 right, -2 left)
        int s = c.state;  // field.py:307-356
        if ((xi == RIGHT_OUT_OF_BOUNDS || yi == RIGHT_OUT_OF_BOUNDS || zi == RIGHT_OUT_OF_BOUNDS) && s < PK_ERROROUTOFBOUNDS) s = PK_ERROROUTOFBOUNDS;
        if (zi == LEFT_OUT_OF_BOUNDS && s < PK_ERRORTHROUGHSURFACE) s = PK_ERRORTHROUGHSURFACE;
        // field.py:359-378: a non-finite barycentric coordinate makes the (wrapped-around) gather NaN, then everything is zeroed
        const bool bad = !(isfinite(xsi) && isfinite(eta) && isfinite(zeta) && isfinite(tau));
        if (bad && s < PK_ERRORINTERPOLATION) s = PK_ERRORINTERPOLATION;
        c.state = s;
        return;
    }
    const bool lenT = tau > 0, lenZ = !(zeta <= 0);
    const int key = (ti << 2) | (lenT ? 2 : 0) | (lenZ ? 1 : 0);
    const uint32_t b00 = ((uint32_t)zi * F.st_z + (uint32_t)yi * F.st_y + (uint32_t)xi) * (uint32_t)sizeof(FT);
    const double omt = 1 - tau, omz = 1 - zeta, omx = 1 - xsi, ome = 1 - eta;
    const double w00 = omx * ome, w01 = xsi * ome, w10 = omx * eta, w11 = xsi * eta;
    double uu = 0.0, vv = 0.0, ww = 0.0;
    const bool bc = !D3 && PK_FAST_BLOCK_CACHE && (fl & FA_BLK) != 0;
    // (in-bounds indices ravel to a non-negative `ei` that names the cell; the memo updates above already dropped a block of another (t, z))
    bool reuse = false;
    if (bc && (bmode & 1)) reuse = __builtin_amdgcn_ballot_w64(c.ei != c.bei) == 0;  // every active lane: same cell, same (t, z)
    if (reuse) {
        const pk_tab2 b0 = T.blk[0], b1 = T.blk[FAST_WG], b2 = T.blk[2 * FAST_WG], b3 = T.blk[3 * FAST_WG];
        uu = w00 * b0.x + w01 * b0.y + w10 * b1.x + w11 * b1.y;
        vv = w00 * b2.x + w01 * b2.y + w10 * b3.x + w11 * b3.y;
    } else {
        const bool keep = bc && (bmode & 2);
        pk_tab2* const blk = keep ? T.blk : nullptr;
        for (bool done = false; !done;) {
            // everything derived from the wave-uniform key is formed BEFORE the lane test: inside `if (key == uk)` the optimiser
            // may substitute the (divergent) key for uk, which would move the level arithmetic back into vector registers
            const int uk = uniform_i32(key);
            const int uti = uk >> 2;
            int s0 = uti, s1 = uti + 1;  // has_ti: 0 <= ti <= nt-2; otherwise ti == 0 and the second level is never read
            if (fl & FA_RING) {          // ring of time levels: level L lives in slot L % nslots
                s0 = (int)((uint32_t)s0 % (uint32_t)F.nslots);
                s1 = (int)((uint32_t)s1 % (uint32_t)F.nslots);
            }
            int64_t o0 = (int64_t)s0 * F.lvl_b, o1 = (int64_t)s1 * F.lvl_b;
            int lens = uk & 3;
            asm volatile("" : "+s"(o0), "+s"(o1), "+s"(lens));  // opaque scalars: no path back to the divergent key
            if (key == uk) {
                if (lens == 3) uvw_fast<FT, D3, true, true>(F, o0, o1, b00, tau, omt, zeta, omz, w00, w01, w10, w11, uu, vv, ww, blk);
                else if (lens == 2) uvw_fast<FT, D3, true, false>(F, o0, o1, b00, tau, omt, zeta, omz, w00, w01, w10, w11, uu, vv, ww, blk);
                else if (lens == 1) uvw_fast<FT, D3, false, true>(F, o0, o1, b00, tau, omt, zeta, omz, w00, w01, w10, w11, uu, vv, ww, blk);
                else uvw_fast<FT, D3, false, false>(F, o0, o1, b00, tau, omt, zeta, omz, w00, w01, w10, w11, uu, vv, ww, blk);
                done = true;
            }
        }
        if (keep) c.bei = c.ei;
    }
    if (fl & FA_SPH) {  // _xinterpolators.py:183-187
        double conv;
        if (PF && pos_f32) conv = (double)((float)F.deg2m * cosf((float)y * DEG2RADF));
        else conv = F.deg2m * cos_lat(y * DEG2RAD);
        if constexpr (PK_FAST_LEAN != 0) {  // u / conv, v / deg2m within 1.5 ulp: 7 operations instead of 22
            uu = uu * rcp_lean(conv);
            vv = vv * F.inv_deg2m;
        } else {
            uu /= conv;
            vv = div_by_recip(vv, F.deg2m, F.inv_deg2m);
        }
    }
    // field.py:373-378 (one unordered compare answers "uu or vv is NaN")
    if (__builtin_expect(__builtin_isunordered(uu, vv) || (D3 && ww != ww), 0)) {
        if (c.state < PK_ERRORINTERPOLATION) c.state = PK_ERRORINTERPOLATION;
    }
    u = uu;
    v = vv;
    w = ww;
}

#ifdef PK_USER_KERNELS
// Field.eval (field.py:145-185) with XLinear for scalar field `slot` of FastA::S at the same evaluation site: the searches, hints, memo
// and state rules of eval_uvw_fast, one gather, no unit conversion.  Only in run-time compiled modules (a user kernel that samples a
// scalar field rides in the dedicated kernel); the arithmetic is xlinear<FT>'s, which the parity tests compare it with at rtol 0.
template <class FT, bool PF>
PK_DEV double eval_scalar_fast(const KArgs& a, const FastTabs& T, FCtx& c, int slot, double t, double z, double y, double x, unsigned it,
                               int klo) {
    const FastA& F = a.fast;
    int ti = 0;
    double tau = 0.0;
    if (F.has_ti) {
        if (!(0 <= t) || !(t <= F.tlen)) {
            c.state = PK_ERROROUTSIDETIMEINTERVAL;
            twe_note(a, it, klo);
            return 0.0;
        }
        if (t != c.mt) {
            int idx;
            fast_search(T.time, F.nt, F.t0, F.t1, t, c.ht, idx, c.mtau);
            c.mt = t;
            c.bei = FAST_NO_BLOCK;
        }
        ti = c.ht;
        tau = c.mtau;
    }
    int zi = 0, yi = 0, xi = 0;
    double zeta = 0.0, eta = 0.0, xsi = 0.0;
    if (F.has_z) {
        if (!(z == c.mz)) {
            fast_search(T.depth, F.gnz, F.z0, F.z1, z, c.hz, c.zi, c.mzeta);
            c.mz = z;
            c.bei = FAST_NO_BLOCK;
        }
        zi = c.zi;
        zeta = c.mzeta;
    }
    if (F.has_y) fast_search(T.lat, F.gny, F.y0, F.y1, y, c.hy, yi, eta);
    if (F.has_x) fast_search(T.lon, F.gnx, F.x0, F.x1, x, c.hx, xi, xsi);
    c.ei = (int32_t)((uint32_t)xi * F.ex + (uint32_t)yi * F.ey + (uint32_t)zi * F.ez);
    if ((xi | yi | zi) < 0) {
        int s = c.state;
        if ((xi == RIGHT_OUT_OF_BOUNDS || yi == RIGHT_OUT_OF_BOUNDS || zi == RIGHT_OUT_OF_BOUNDS) && s < PK_ERROROUTOFBOUNDS) s = PK_ERROROUTOFBOUNDS;
        if (zi == LEFT_OUT_OF_BOUNDS && s < PK_ERRORTHROUGHSURFACE) s = PK_ERRORTHROUGHSURFACE;
        const bool bad = !(isfinite(xsi) && isfinite(eta) && isfinite(zeta) && isfinite(tau));
        if (bad && s < PK_ERRORINTERPOLATION) s = PK_ERRORINTERPOLATION;
        c.state = s;
        return 0.0;
    }
    const bool lenT = tau > 0, lenZ = !(zeta <= 0);
    const uint32_t b00 = ((uint32_t)zi * F.st_z + (uint32_t)yi * F.st_y + (uint32_t)xi) * (uint32_t)sizeof(FT);
    const double omt = 1 - tau, omz = 1 - zeta, omx = 1 - xsi, ome = 1 - eta;
    const double w00 = omx * ome, w01 = xsi * ome, w10 = omx * eta, w11 = xsi * eta;
    int s0 = ti, s1 = ti + 1;
    if (F.nslots < F.nt) {
        s0 = (int)((uint32_t)s0 % (uint32_t)F.nslots);
        s1 = (int)((uint32_t)s1 % (uint32_t)F.nslots);
    }
    const char* base = F.S[slot];
    const char* l0 = base + (int64_t)s0 * F.lvl_b;
    const char* l1 = base + (int64_t)s1 * F.lvl_b;
    Rows r;
    double val;
    if (lenT && lenZ) { load_rows<FT, true, true>(r, l0, l1, b00, F.dyb, F.dzb); val = interp_rows<true, true>(r, tau, omt, zeta, omz, w00, w01, w10, w11); }
    else if (lenT) { load_rows<FT, true, false>(r, l0, l1, b00, F.dyb, F.dzb); val = interp_rows<true, false>(r, tau, omt, zeta, omz, w00, w01, w10, w11); }
    else if (lenZ) { load_rows<FT, false, true>(r, l0, l1, b00, F.dyb, F.dzb); val = interp_rows<false, true>(r, tau, omt, zeta, omz, w00, w01, w10, w11); }
    else { load_rows<FT, false, false>(r, l0, l1, b00, F.dyb, F.dzb); val = interp_rows<false, false>(r, tau, omt, zeta, omz, w00, w01, w10, w11); }
    if (__builtin_expect(val != val, 0)) {
        if (c.state < PK_ERRORINTERPOLATION) c.state = PK_ERRORINTERPOLATION;
    }
    return val;
}
#endif

}  // namespace pk
