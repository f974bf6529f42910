// pk_fast_cgrid.h -- VectorField.eval for BASELINE configurations 3 / 4: CGrid_Velocity on a spherical curvilinear C-grid with
// float64 node coordinates (field.py:250-304, index_search.py:94-295, spatialhash.py:389-535, _xinterpolators.py:193-332).
//
// Same arithmetic, expression by expression, as the general eval_uvw<FT, 1, 1, false> of pk_device.h (the parity tests run both
// against the same fixtures, and against each other at rtol 0), laid out for what bounds that program on MI355X -- instruction
// issue and memory latency, not bandwidth:
//   * everything the general path re-derives per evaluation from the grid / field descriptors is folded into one compact
//     wave-uniform `FastC` block by the host (pk_api.hip: fill_fastc): element strides in bytes, the six staggered offsets of a
//     cell, ravel strides;
//   * a second per-cell table (`ct2`, 256 B per cell = two cache lines) holds, besides the tangent-plane basis, the
//     query-independent part of the bilinear inverse (index_search.py:122-177: the coefficients a0..a3, b1, b3, the products
//     4*aa, and the prefixes of bb and cc that do not contain the query point -- formed in the reference's evaluation order, so the
//     remaining operations give the same bits) and the antimeridian-unwrapped corner longitudes of CGrid_Velocity
//     (_xinterpolators.py:230-233);
//   * a lane caches that record and the 12 raw staggered field values of its cell: the 15 rows of the point-in-cell test and the field
//     values in its LDS slot, the 8 corner coordinates and the tags of the slot in registers (CM below: the split that gives 3 waves per
//     SIMD); a miss fetches record AND field values of the new cell in ONE memory round trip (the general path needs two dependent
//     ones: the record, then -- after the point-in-cell test -- the values);
//   * phi2D_lin rows with a literal 0 / 1 argument are written with their exact zero products removed, the cosine of the
//     particle latitude is shared between the unit-sphere query point and the metres -> degrees conversion, the ring slot
//     of a time level (`level % nslots`) is computed on the scalar unit once per distinct level of a wavefront.
// The rare branches (no valid guess, neighbour probe rejected, degenerate bilinear inverse) call the general device functions.
#pragma once
#include "pk_device.h"
#include "pk_fast_agrid.h"

namespace pk {

// ct2 record (doubles): 0..2 eu, 3..5 ev, 6 a0, 7 a1, 8 a2, 9 a3, 10 b1, 11 b3, 12 4*aa, 13 bb0, 14 cc0, 15 quantised box,
// 16..19 unwrapped corner longitudes, 20..23 corner latitudes, 24..27 pv (projected corners, second coordinate: only the degenerate
// branch of the bilinear inverse reads them), 28..31 unused
constexpr int CT2_STRIDE = 32;
// What a lane caches of its cell, and where (`CM`, a template parameter of everything below; one value per kernel, pk_kernels.h).
// The LDS slot and the register count together bound the occupancy of these one-wavefront workgroups: 23 record rows + 12 field values
// in LDS are 232 B per lane = 10 workgroups per CU (160 KB), and ~170 VGPRs are 2 waves per SIMD.  Bit 0 (CG_FV_REGS): the 12 staggered
// field values live in registers; bit 1 (CG_PXY_REGS): the 8 corner coordinates CGrid_Velocity reads (rows 16..23 of the record) do.
// Both are read with compile-time indices only.  The 15 rows of the point-in-cell test always stay in LDS.
// Bit 2 (CG_PXY_GLOBAL): the corner coordinates are not cached at all -- the velocity sample reads them from the cell's record in the
// table (rows 16..23, one more cache line of a record whose first lines the search just read).  For AdvectionDiffusionM1's program, where
// six of the seven samples of a step are scalar samples that never look at them: 16 VGPRs less to carry around every one of them, and a
// cell change fetches 128 B of the record instead of 192.
// Bit 3 (CG_DMA, round 6): a cell change moves the record and -- for float32 fields -- the staggered field values from memory STRAIGHT
// into the lane's LDS slot (`global_load_lds_dwordx4` / `_dword`: the LDS-DMA path of gfx950) instead of through registers.  Rounds 2-5
// staged the 24 doubles of a record in 48 VGPRs (+ 12 field values) between the loads and the 23 + 12 `ds_write`s -- the register peak of
// the whole evaluation, at the 168-VGPR ceiling of 3 waves per SIMD: 160 B / lane of scratch in the RK45 kernel, whose spill write-back
// was 12 % of its HBM traffic.  The DMA lands lane L's 16 bytes of pair k at `base + k * 1024 + L * 16`, so the slot is laid out in PAIRS
// of rows, [pair][lane][2] (the test reads them back with 16-byte LDS reads), and row 15 -- the quantised box -- lives there too; the
// field values keep their [k][lane] layout (4-byte DMA).  Same values from the same addresses: same bits.  Lanes outside the exec mask
// neither load nor write, so the slots of the lanes that stayed in their cell are untouched.
// Bit 4 (CG_PIN): the scalars of FastC that every cell change reads are pinned in scalar registers (cg_pin).  Bit 5 (CG_NO_TMEMO): no memo
// of the last time searched -- AdvectionRK45's six stage times are all different, the memo never hits there and costs two registers.
constexpr int CG_FV_REGS = 1, CG_PXY_REGS = 2, CG_PXY_GLOBAL = 4, CG_DMA = 8, CG_PIN = 16, CG_NO_TMEMO = 32;
constexpr int fc_rec_rows(int cm) {  // LDS rows of a lane's slot: record rows 0..14 (then 16..23); CG_DMA: whole pairs, rows 0..15 (then 16..23)
    return (cm & CG_DMA) ? ((cm & (CG_PXY_REGS | CG_PXY_GLOBAL)) ? 16 : 24) : ((cm & (CG_PXY_REGS | CG_PXY_GLOBAL)) ? 15 : 23);
}
constexpr int fc_fv_lds(int cm) { return (cm & CG_FV_REGS) ? 0 : 12; }       // field values of a lane kept in LDS
constexpr int FC_LANES = 64;     // one-wavefront workgroups (see CC_LANES)
#ifndef PK_CG_HOPS
#define PK_CG_HOPS 0  // AdvectionRK45, after the guessed cell rejected the point: 0 = ONE probe of the cell floor(xsi, eta) cells away (its stages
                      // with a dt of hours land a cell or two away, and on a smooth mesh the bilinear inverse of the guessed cell extrapolates
                      // that far), 1 = the adjacent cell like every other kernel, 3 = up to three hops.  BASELINE config 5, same box
                      // (profiles/r04_d_c5_variants_ab.txt, r04_e_c5_variants_ab.txt): 14.6 ms (1), 13.4 ms (0), 15.3 ms (a loop of up to
                      // three hops: its registers cost more than the table walks it saves -- removed, like an affine predictor of the first
                      // cell from the corner coordinates: 15.1 vs 13.5 ms, profiles/r04_p_rk45_predictor_ab.txt)
#endif

struct CgLds {
    const pk_tab2* time;   // {a, 1/width} tables
    const pk_tab2* depth;
    double* rec;           // [fc_rec_rows(CM)][64] + lane      (CG_DMA: [pairs][64][2] + 2 * lane)
    void* fv;              // [fc_fv_lds(CM)][64] of the field dtype + lane
    char* rec_w;           // wave-uniform bases of the same two regions: where the LDS-DMA of CG_DMA lands (it adds lane * size itself)
    char* fv_w;
    // Scalars of FastC that every cell change / evaluation reads, PINNED in scalar registers by the kernels (cg_pin; see FastTabs::tlen of
    // pk_fast_agrid.h): round 5's RK45 kernel re-loaded {st_z, st_y, cb, lvl_b} and the table pointer from the kernel-argument segment at
    // every fetch site -- 14.5 scalar loads per wave-evaluation (profiles/r05f_c5_rk45_pmc.md), each a ~200-cycle `s_waitcnt lgkmcnt(0)`
    int32_t st_z, st_y, cb;
    int64_t lvl_b;
    const double* ct2;
    double tlen;
};
template <int CM>
PK_DEV void cg_pin(CgLds& L, const FastC& F) {
    L.st_z = F.st_z;
    L.st_y = F.st_y;
    L.cb = F.cb;
    L.lvl_b = F.lvl_b;
    L.ct2 = F.ct2;
    L.tlen = F.tlen;
    // (measured on BASELINE configs 3 / 5, profiles/r06c_*: the pin halves the scalar loads per wave-evaluation, 13.0 -> 6.9, and is
    // worth nothing in time -- within +-1 % everywhere, slightly negative for AdvectionRK4_3D and M1; kept for the RK45 kernel, where
    // together with CG_PXY_GLOBAL | CG_DMA it brings the scratch from 160 to 60 B / lane)
    if constexpr ((CM & CG_PIN) != 0) asm volatile("" : "+s"(L.st_z), "+s"(L.st_y), "+s"(L.cb), "+s"(L.lvl_b), "+s"(L.ct2), "+s"(L.tlen));
}
// row k (compile-time) of the record in the lane's slot
template <int CM>
PK_DEV double cg_rec_row(const double* rec, int k) {
    if constexpr ((CM & CG_DMA) != 0) return rec[(k >> 1) * (2 * 64) + (k & 1)];
    else return rec[k * 64];
}
PK_DEV void cg_dma16(const void* g, char* lds_w) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_w, 16, 0, 0);
}
PK_DEV void cg_dma4(const void* g, char* lds_w) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_w, 4, 0, 0);
}
// per-particle evaluation context of the fast C-grid kernels (FT: dtype of the velocity fields)
template <class FT, int CM>
struct CCtxT {
    int state;
    int32_t ei;
    int ht, hz;          // cells of the previous time / depth search (valid cell indices: hints)
    int zi;              // memo: index or out-of-bounds code of the last depth searched
    int gy, gx;          // the guess of the next horizontal search: the cell the previous one found (index_search.py:269-274)
    int rc_cell;         // cell whose record sits in the lane's LDS slot, -1 = none
    int fv_cell, fv_zt;  // tags of the cached field values: cell, (ti << 13) | (zi << 1) | (level ti+1 cached)
    double mt, mtau, mz, mzeta;
    // AdvectionDiffusionM1's program: five of the seven samples of a step are taken at the particle's own latitude, five at its own
    // longitude -- the kernel computes their sines / cosines once per step (cg_home_sincos) and a sample whose coordinate is bitwise the
    // particle's takes them from here (unused, and optimised away, elsewhere)
    double q_sl, q_cl, q_so, q_co;
    FT fvr[(CM & CG_FV_REGS) ? 12 : 1];      // the cached field values (CG_FV_REGS)
    double pxy[(CM & CG_PXY_REGS) ? 8 : 1];  // unwrapped corner longitudes, corner latitudes of rc_cell (CG_PXY_REGS)
};
template <class FT, int CM>
PK_DEV void cctx_init(CCtxT<FT, CM>& c, int state, int32_t ei, int gy, int gx) {
    c.state = state;
    c.ei = ei;
    c.ht = c.hz = 0;
    c.zi = 0;
    c.gy = gy;
    c.gx = gx;
    c.rc_cell = -1;
    c.fv_cell = -1;
    c.fv_zt = 0;
    c.mt = c.mz = __builtin_nan("");
    c.mtau = c.mzeta = 0.0;
    c.q_sl = c.q_cl = c.q_so = c.q_co = 0.0;
#pragma unroll
    for (int k = 0; k < ((CM & CG_FV_REGS) ? 12 : 1); k++) c.fvr[k] = (FT)0;
#pragma unroll
    for (int k = 0; k < ((CM & CG_PXY_REGS) ? 8 : 1); k++) c.pxy[k] = 0.0;
}

// Byte offsets (b0, b1) of cell element `e` inside the ring slots of levels ti and ti+1 (slot_off of pk_device.h).  A ring needs
// `level % nslots`: the level is wave-uniform unless particles of one wavefront sit on different levels, so the modulo runs on the
// scalar unit, once per distinct level of the wavefront (readfirstlane waterfall); every lane keeps the offsets of ITS level.
PK_DEV void cg_level_offsets(const FastC& F, const CgLds& L, uint32_t e, int ti, int64_t& b0, int64_t& b1) {
    const int64_t vb = (int64_t)((uint64_t)e * (uint64_t)(uint32_t)L.cb);
    int64_t o0 = 0, o1 = 0;
    for (bool done = false; !done;) {
        const int uti = uniform_i32(ti);
        int s0 = uti, s1 = mini(uti + 1, F.nt - 1);
        if (F.nslots < F.nt) {  // level L lives in slot L % nslots
            s0 = (int)((uint32_t)s0 % (uint32_t)F.nslots);
            s1 = (int)((uint32_t)s1 % (uint32_t)F.nslots);
        }
        const int64_t u0 = (int64_t)s0 * L.lvl_b, u1 = (int64_t)s1 * L.lvl_b;  // derived from the uniform level BEFORE the lane test
        if (ti == uti) {
            o0 = u0;
            o1 = u1;
            done = true;
        }
    }
    b0 = vb + o0;
    b1 = vb + o1;
}
// Fetch the ct2 record of `cell` into the lane's LDS slot -- and, WITH_F, the staggered field values of (zi, yi, xi) at level ti
// (and ti+1 if lenT) in the same memory round trip.  Returns the packed quantised box of the cell (row 15).
template <class FT, bool D3>
PK_DEV void cg_issue_fields(const FastC& F, const CgLds& L, int zi, int yi, int xi, int ti, bool lenT, FT raw[12]) {
    const uint32_t e = (uint32_t)zi * (uint32_t)L.st_z + (uint32_t)yi * (uint32_t)L.st_y + (uint32_t)xi;  // < 2^31 elements per level (host check)
    if constexpr (!D3) {
        // (F.vp: wave-uniform) the cell-packed pair copy: both levels of the cell in one 8-value group.
        // pair L holds levels (L, L + 1); a sample exactly on the highest resident level (tau == 0: !lenT) has no pair of its own
        // and reads the upper half of the pair below.  A sample BETWEEN the highest resident level and the next one cannot exist (the
        // window pause of the step loop keeps it out); should that invariant ever break, the lane reads the level rings below instead
        // of the bytes behind its group.
        if (F.vp && !(ti >= (int)F.vp_hi && lenT)) {
            const bool last = ti >= (int)F.vp_hi;
            const int pair = last ? ti - 1 : ti;
            int64_t off = 0;
            for (bool done = false; !done;) {
                const int up = uniform_i32(pair);
                const int64_t u = (int64_t)(F.nslots < F.nt ? (int)((uint32_t)up % (uint32_t)F.nslots) : up) * F.vp_slot_b;
                if (pair == up) {
                    off = u;
                    done = true;
                }
            }
            const FT* g = reinterpret_cast<const FT*>(F.vp + off) + (int64_t)e * 8 + (last ? 4 : 0);
            if constexpr (sizeof(FT) == 4) {
                const pk_float4 a = *reinterpret_cast<const pk_float4*>(g);
                raw[0] = a.x; raw[1] = a.y; raw[2] = a.z; raw[3] = a.w;
            } else {
                const pk_double2 a = *reinterpret_cast<const pk_double2*>(g), b = *reinterpret_cast<const pk_double2*>(g + 2);
                raw[0] = a.x; raw[1] = a.y; raw[2] = b.x; raw[3] = b.y;
            }
            raw[4] = raw[5] = (FT)0;
#pragma unroll
            for (int k = 6; k < 12; k++) raw[k] = (FT)0;
            if (lenT) {
                if constexpr (sizeof(FT) == 4) {
                    const pk_float4 a = *reinterpret_cast<const pk_float4*>(g + 4);
                    raw[6] = a.x; raw[7] = a.y; raw[8] = a.z; raw[9] = a.w;
                } else {
                    const pk_double2 a = *reinterpret_cast<const pk_double2*>(g + 4), b = *reinterpret_cast<const pk_double2*>(g + 6);
                    raw[6] = a.x; raw[7] = a.y; raw[8] = b.x; raw[9] = b.y;
                }
            }
            return;
        }
    }
    int64_t b0, b1;
    cg_level_offsets(F, L, e, ti, b0, b1);
    raw[0] = *reinterpret_cast<const FT*>(F.U + F.dU0 + b0);
    raw[1] = *reinterpret_cast<const FT*>(F.U + F.dU1 + b0);
    raw[2] = *reinterpret_cast<const FT*>(F.V + F.dV0 + b0);
    raw[3] = *reinterpret_cast<const FT*>(F.V + F.dV1 + b0);
    raw[4] = D3 ? *reinterpret_cast<const FT*>(F.W + F.dW0 + b0) : (FT)0;
    raw[5] = D3 ? *reinterpret_cast<const FT*>(F.W + F.dW1 + b0) : (FT)0;
#pragma unroll
    for (int k = 6; k < 12; k++) raw[k] = (FT)0;
    if (lenT) {  // per lane: a particle exactly on a time level reads that level only
        raw[6] = *reinterpret_cast<const FT*>(F.U + F.dU0 + b1);
        raw[7] = *reinterpret_cast<const FT*>(F.U + F.dU1 + b1);
        raw[8] = *reinterpret_cast<const FT*>(F.V + F.dV0 + b1);
        raw[9] = *reinterpret_cast<const FT*>(F.V + F.dV1 + b1);
        if (D3) {
            raw[10] = *reinterpret_cast<const FT*>(F.W + F.dW0 + b1);
            raw[11] = *reinterpret_cast<const FT*>(F.W + F.dW1 + b1);
        }
    }
}
// The same values by LDS-DMA (CG_DMA, 4-byte field dtypes): value k of lane L lands at fv_w + k * 256 + L * 4 -- the [k][lane] layout the
// interpolation reads.  Slots the register path zero-fills (W in the 2-D kernels, level ti+1 of a sample ON a level) stay as they are:
// the readers never use them (eval_uvw_cgrid reads 6..11 only if lenT, and W only if D3).
template <class FT, bool D3>
PK_DEV void cg_issue_fields_dma(const FastC& F, const CgLds& L, int zi, int yi, int xi, int ti, bool lenT) {
    static_assert(sizeof(FT) == 4, "LDS-DMA moves 4, 12 or 16 bytes per lane");
    char* const fv_w = L.fv_w;
    const uint32_t e = (uint32_t)zi * (uint32_t)L.st_z + (uint32_t)yi * (uint32_t)L.st_y + (uint32_t)xi;
    int64_t b0, b1;
    cg_level_offsets(F, L, e, ti, b0, b1);
    cg_dma4(F.U + F.dU0 + b0, fv_w + 0 * 256);
    cg_dma4(F.U + F.dU1 + b0, fv_w + 1 * 256);
    cg_dma4(F.V + F.dV0 + b0, fv_w + 2 * 256);
    cg_dma4(F.V + F.dV1 + b0, fv_w + 3 * 256);
    if (D3) {
        cg_dma4(F.W + F.dW0 + b0, fv_w + 4 * 256);
        cg_dma4(F.W + F.dW1 + b0, fv_w + 5 * 256);
    }
    if (lenT) {
        cg_dma4(F.U + F.dU0 + b1, fv_w + 6 * 256);
        cg_dma4(F.U + F.dU1 + b1, fv_w + 7 * 256);
        cg_dma4(F.V + F.dV0 + b1, fv_w + 8 * 256);
        cg_dma4(F.V + F.dV1 + b1, fv_w + 9 * 256);
        if (D3) {
            cg_dma4(F.W + F.dW0 + b1, fv_w + 10 * 256);
            cg_dma4(F.W + F.dW1 + b1, fv_w + 11 * 256);
        }
    }
}
template <class FT, int CM>
PK_DEV void cg_store_fields(CCtxT<FT, CM>& c, const CgLds& L, int cell, int zi, int ti, bool lenT, const FT raw[12]) {
    if constexpr ((CM & CG_FV_REGS) != 0) {
#pragma unroll
        for (int k = 0; k < 12; k++) c.fvr[k] = raw[k];
    } else {
        FT* fv = (FT*)L.fv;
#pragma unroll
        for (int k = 0; k < 12; k++) fv[k * FC_LANES] = raw[k];
    }
    c.fv_cell = cell;
    c.fv_zt = (ti << 13) | (zi << 1) | (lenT ? 1 : 0);
}
template <class FT, int CM>
PK_DEV bool cg_fields_cached(const CCtxT<FT, CM>& c, int cell, int zi, int ti, bool lenT) {
    return c.fv_cell == cell && (c.fv_zt >> 1) == ((ti << 12) | zi) && (!lenT || (c.fv_zt & 1));
}

// the staggered field values of (zi, yi, xi) at level ti (and ti+1 if lenT) into the lane's cache, by whichever path the kernel uses;
// WAIT: the caller reads them next (the DMA needs its own vmcnt wait -- nothing else orders an LDS read behind it)
template <class FT, bool D3, int CM, bool WAIT>
PK_DEV void cg_refresh_fields(const FastC& F, const CgLds& L, CCtxT<FT, CM>& c, int cell, int zi, int yi, int xi, int ti, bool lenT) {
    if constexpr ((CM & CG_DMA) != 0 && !(CM & CG_FV_REGS) && sizeof(FT) == 4) {
        if (D3 || !F.vp) {  // (F.vp: wave-uniform; the opt-in pair copies keep the register path)
            cg_issue_fields_dma<FT, D3>(F, L, zi, yi, xi, ti, lenT);
            if (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            c.fv_cell = cell;
            c.fv_zt = (ti << 13) | (zi << 1) | (lenT ? 1 : 0);
            return;
        }
    }
    FT raw[12];
    cg_issue_fields<FT, D3>(F, L, zi, yi, xi, ti, lenT, raw);
    cg_store_fields<FT, CM>(c, L, cell, zi, ti, lenT, raw);
}

template <class FT, bool D3, bool WITH_F, int CM>
PK_DEV double cg_fetch_cell(const FastC& F, const CgLds& L, CCtxT<FT, CM>& c, int cell, int yi, int xi, int zi, int ti, bool lenT) {
    const double* g = L.ct2 + (int64_t)cell * CT2_STRIDE;
    if constexpr ((CM & CG_DMA) != 0) {
        // record rows 0..15 (and 16..23 unless they live in registers / stay in the table) as 16-byte pairs straight into the slot
        constexpr int NP = fc_rec_rows(CM) / 2;
#pragma unroll
        for (int k = 0; k < 8; k++) cg_dma16(g + 2 * k, L.rec_w + k * 1024);
        if constexpr (NP == 12) {
#pragma unroll
            for (int k = 8; k < 12; k++) cg_dma16(g + 2 * k, L.rec_w + k * 1024);
        }
        if constexpr ((CM & CG_PXY_REGS) != 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) ldpair(g + 16 + 2 * k, c.pxy[2 * k], c.pxy[2 * k + 1]);
        }
        if (WITH_F) {
            const bool wantf = zi >= 0 && !cg_fields_cached(c, cell, zi, ti, lenT);
            if (wantf) cg_refresh_fields<FT, D3, CM, false>(F, L, c, cell, zi, yi, xi, ti, lenT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        c.rc_cell = cell;
        return cg_rec_row<CM>(L.rec, 15);
    } else {
        constexpr int NR = (CM & CG_PXY_GLOBAL) ? 16 : 24;  // rows of the record a lane keeps
        double r[NR];
#pragma unroll
        for (int k = 0; k < NR / 2; k++) ldpair(g + 2 * k, r[2 * k], r[2 * k + 1]);
        FT raw[12];
        const bool wantf = WITH_F && zi >= 0 && !cg_fields_cached(c, cell, zi, ti, lenT);
        if (WITH_F) {
            if (wantf) cg_issue_fields<FT, D3>(F, L, zi, yi, xi, ti, lenT, raw);
        }
        double* rec = L.rec;
#pragma unroll
        for (int k = 0; k < 15; k++) rec[k * FC_LANES] = r[k];
        if constexpr ((CM & CG_PXY_GLOBAL) != 0) {
            // (read where they are used)
        } else if constexpr ((CM & CG_PXY_REGS) != 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) c.pxy[k] = r[16 + k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) rec[(15 + k) * FC_LANES] = r[16 + k];
        }
        c.rc_cell = cell;
        if (WITH_F) {
            if (wantf) cg_store_fields<FT, CM>(c, L, cell, zi, ti, lenT, raw);
        }
        return r[15];
    }
}

// Sines / cosines NEAR an angle whose sine and cosine are known (PK_CG_NEAR, round 6): a sample point of a step lies within a fraction of a
// degree of the particle's own position, and the four edge points of CGrid_Velocity's geodetic distances within a cell of the sample's
// latitude, so sin / cos(a0 + d) = s0 cos d + c0 sin d / c0 cos d - s0 sin d with the Taylor polynomials of sin d and cos d - 1 for
// |d| <= 2^-7 rad (0.45 degrees: truncation d^7 / 5040 < 3.5e-19, d^8 / 40320 < 4e-22) -- 13 fp64 operations instead of the 50-odd of a
// reduction + two minimax kernels, no branch.  Error: that of (s0, c0) (< 1 ulp, sincos_geo) + one rounding: < 2.5 ulp, where the
// reference's own libm promises <= 1 ulp: trajectories stay within the 1e-12 every parity test states, but
// are no longer the bits of the general program, which has no "own position" to start from.  Larger |d|: the full routines.
#ifndef PK_CG_LEAN
#define PK_CG_LEAN 1
#endif
#if PK_CG_LEAN
// With PK_CG_LEAN the arithmetic of the evaluation below (query point, edge points, time lerp of the staggered values, Jacobian, the weighted
// sums of CGrid_Velocity, XLinear of the scalar samples) may be CONTRACTED: a * b + c as one fused operation, one rounding instead of NumPy's
// two -- never further from the exact value.  The translation units are compiled with -ffp-contract=off (the general programs reproduce the
// two roundings); the pragma opens these functions only.  RK4_3D -3.2 %, M1 -1.6 %, RK45 -1 % (profiles/r06v_contraction_ab.txt).
#define PK_CG_FMA _Pragma("clang fp contract(fast)")
#else
#define PK_CG_FMA
#endif
#ifndef PK_CG_NEAR
#define PK_CG_NEAR 3  // bit 0: the sample point from the particle's own position; bit 1: the edge points from the sample point
#endif
#ifndef PK_CG_NEAR_RK45
#define PK_CG_NEAR_RK45 PK_CG_NEAR
#endif
#ifndef PK_CG_NEAR_M1
#define PK_CG_NEAR_M1 PK_CG_NEAR
#endif
static constexpr double CG_NEAR_MAX = 0.0078125;
PK_DEV void sincos_near(double d, double s0, double c0, double& s, double& c) {
    const double z = d * d;
    const double sd = fma(d * z, fma(z, 8.33333333333333333e-03, -1.66666666666666667e-01), d);
    const double cm = z * fma(z, fma(z, -1.38888888888888889e-03, 4.16666666666666667e-02), -0.5);
    s = s0 + fma(c0, sd, s0 * cm);
    c = c0 + fma(-s0, sd, c0 * cm);
}
PK_DEV double cos_near(double d, double s0, double c0) {
    const double z = d * d;
    const double sd = fma(d * z, fma(z, 8.33333333333333333e-03, -1.66666666666666667e-01), d);
    const double cm = z * fma(z, fma(z, -1.38888888888888889e-03, 4.16666666666666667e-02), -0.5);
    return c0 + fma(-s0, sd, c0 * cm);
}

template <class FT, int CM>
PK_DEV void cg_home_sincos(CCtxT<FT, CM>& c, double y, double x) {
    sincos_geo(y * DEG2RAD, c.q_sl, c.q_cl);
    sincos_geo(x * DEG2RAD, c.q_so, c.q_co);
}

// curvilinear_point_in_cell (index_search.py:94-177) on the record in the lane's LDS slot.  The operations of point_in_cell ->
// spherical_project_query -> bilinear_inverse of pk_device.h in their order (the cell-only sub-expressions were formed by the table
// build in the same order); with PK_CG_LEAN = 0 bit for bit their result, with the lean square root and quotients the same bits in
// practice (0 ulp from the library's over 1e8 operands, tools/lean_math_check.hip).  `cell`: for the degenerate branch (reads pv from the
// global record).
template <class Row>
PK_DEV bool cg_point_in_cell_rows(const FastC& F, Row row, int cell, double qX, double qY, double qZ, double& xsi, double& eta) {
    // (NOT contracted: bb * bb - 4 aa cc cancels on near-parallelogram cells, and which cells the host flags as ill-conditioned --
    // illconditioned_cell_kernel, neighbour probing off -- is decided with the reference's two roundings)
    const double eu0 = row(0), eu1 = row(1), eu2 = row(2);
    const double ev0 = row(3), ev1 = row(4), ev2 = row(5);
    const double a0 = row(6), a1 = row(7), a2 = row(8), a3 = row(9);
    const double b1 = row(10), b3 = row(11);
    const double aa4 = row(12), bb0 = row(13), cc0 = row(14);
    const double xq = qX * eu0 + qY * eu1 + qZ * eu2;  // spherical_project_query
    const double yq = qX * ev0 + qY * ev1 + qZ * ev2;
    const double bb = bb0 + xq * b3 - yq * a3;
    const double cc = cc0 + xq * b1 - yq * a1;
    const double det2 = bb * bb - aa4 * cc;  // 4 * aa * cc with 4 * aa from the table (exact scaling)
    const double det = det2 > 0 ? (PK_CG_LEAN ? sqrt_lean(det2) : sqrt(det2)) : -1.0;
    double e;
    if (__builtin_expect(fabs(aa4) < 4 * 1e-12, 0)) {  // |aa| < 1e-12
        double cn = cc;
        asm volatile("" : "+v"(cn));  // keep the rare branch a branch (see div_by_recip)
        e = -cn / bb;
    } else {
        e = det2 > 0 ? (PK_CG_LEAN ? div_lean(-bb + det, aa4 * 0.5) : (-bb + det) / (aa4 * 0.5)) : -1.0;  // 2 * aa
    }
    const double den = a1 + a3 * e;
    double x;
    if (__builtin_expect(fabs(den) < 1e-12, 0)) {
        const double* g = F.ct2 + (int64_t)cell * CT2_STRIDE;
        double py0, py1, py2, py3;
        ldpair(g + 24, py0, py1);
        ldpair(g + 26, py2, py3);
        x = ((yq - py0) / (py1 - py0) + (yq - py3) / (py2 - py3)) * 0.5;
    } else {
        x = PK_CG_LEAN ? div_lean(xq - a0 - a2 * e, den) : (xq - a0 - a2 * e) / den;
    }
    xsi = x;
    eta = e;
    return (x >= 0) && (x <= 1) && (e >= 0) && (e <= 1);
}
template <int CM = 0>
PK_DEV bool cg_point_in_cell(const FastC& F, const double* rec, int cell, double qX, double qY, double qZ, double& xsi, double& eta) {
    return cg_point_in_cell_rows(F, [rec](int k) { return cg_rec_row<CM>(rec, k); }, cell, qX, qY, qZ, xsi, eta);
}

// XLinear.interp (_xinterpolators.py:112-153) of scalar field `k` (FastC::kh) at a grid position: xlinear<FT> of pk_device.h for a
// float64-coordinate grid, with the field's own strides (an axis the field lacks has stride 0: the two "levels" are then the same
// values, and c * (1 - zeta) + c * zeta is still formed, like the reference does)
template <class FT>
PK_DEV double cg_scalar_xlinear(const FastC& F, int k, int ti, double tau, int zi, double zeta, int yi, int xi, double xsi, double eta) {
    PK_CG_FMA
    const bool lenT = tau > 0, lenZ = !(zeta <= 0);
    if (F.kh_nt[k] == 1 && F.kh_nz[k] == 1) {
        // a field without time and depth axes (the 2-D Kh fields of BASELINE config 5): both "depth levels" are the same two rows -- the
        // same operations on the same values as below (c * (1 - zeta) + c * zeta is still formed), without the level / plane addressing
        // and with two row loads instead of four: M1 43.3 -> 38.9 ms on config 5 (profiles/r04_m_kh2d_rk45g_c3g_ab.txt)
        const char* d0 = F.kh[k];
        const int64_t oy0 = (int64_t)yi * F.kh_sy[k], oy1 = (int64_t)mini(yi + 1, F.kh_ny[k] - 1) * F.kh_sy[k];
        const int64_t ox = (int64_t)xi * (int64_t)sizeof(FT);
        double c00, c01, c10, c11;
        ldpair(reinterpret_cast<const FT*>(d0 + oy0 + ox), c00, c01);
        ldpair(reinterpret_cast<const FT*>(d0 + oy1 + ox), c10, c11);
        if (lenZ) {
            c00 = c00 * (1 - zeta) + c00 * zeta; c01 = c01 * (1 - zeta) + c01 * zeta;
            c10 = c10 * (1 - zeta) + c10 * zeta; c11 = c11 * (1 - zeta) + c11 * zeta;
        }
        return (1 - xsi) * (1 - eta) * c00 + xsi * (1 - eta) * c01 + (1 - xsi) * eta * c10 + xsi * eta * c11;
    }
    const int nt = F.kh_nt[k], ns = F.kh_nslots[k];
    int s0 = ti, s1 = mini(ti + 1, nt - 1);
    if (ns < nt) { s0 = (int)((uint32_t)s0 % (uint32_t)ns); s1 = (int)((uint32_t)s1 % (uint32_t)ns); }
    const char* d0 = F.kh[k] + (int64_t)s0 * F.kh_st[k];
    const char* d1 = F.kh[k] + (int64_t)s1 * F.kh_st[k];
    const int64_t oz0 = (int64_t)zi * F.kh_sz[k], oz1 = (int64_t)mini(zi + 1, F.kh_nz[k] - 1) * F.kh_sz[k];
    const int64_t oy0 = (int64_t)yi * F.kh_sy[k], oy1 = (int64_t)mini(yi + 1, F.kh_ny[k] - 1) * F.kh_sy[k];
    const int64_t ox = (int64_t)xi * (int64_t)sizeof(FT);  // xi + 1 <= nx - 1 for a cell of the grid the field lives on
    double c[2][2];
#pragma unroll
    for (int iz = 0; iz < 2; iz++) {
        if (iz == 1 && !lenZ) break;
        const int64_t oz = iz ? oz1 : oz0;
        double lv[2][2];
#pragma unroll
        for (int iy = 0; iy < 2; iy++) {
            const int64_t o = oz + (iy ? oy1 : oy0) + ox;
            double a0, a1;
            ldpair(reinterpret_cast<const FT*>(d0 + o), a0, a1);
            if (lenT) {
                double b0, b1;
                ldpair(reinterpret_cast<const FT*>(d1 + o), b0, b1);
                a0 = a0 * (1 - tau) + b0 * tau;
                a1 = a1 * (1 - tau) + b1 * tau;
            }
            lv[iy][0] = a0;
            lv[iy][1] = a1;
        }
#pragma unroll
        for (int iy = 0; iy < 2; iy++)
#pragma unroll
            for (int ix = 0; ix < 2; ix++) c[iy][ix] = iz ? c[iy][ix] * (1 - zeta) + lv[iy][ix] * zeta : lv[iy][ix];
    }
    return (1 - xsi) * (1 - eta) * c[0][0] + xsi * (1 - eta) * c[0][1] + (1 - xsi) * eta * c[1][0] + xsi * eta * c[1][1];
}

// VectorField.eval (field.py:250-304) + XGrid.search (xgrid.py:316-356) + CGrid_Velocity.interp (_xinterpolators.py:193-332).
// PF: the sample point may come straight from float32 particle storage (pos_f32); D3: sample W as well.
// WITH_SCALAR (AdvectionDiffusionM1's program): `sk` >= 0 asks for Field.eval (field.py:145-195) of scalar field FastC::kh[sk] instead --
// same search on the same grid (the `ei` guess chain of the particle runs through velocity and scalar samples alike), XLinear on the
// field's nodes; the value is returned in u.  One call site serves all seven samples of a step (sk is a run-time value there).
template <class FT, bool PF, bool D3, bool WITH_SCALAR = false, int CM = 0, int HOPS = 1, int NEAR = PK_CG_NEAR>
PK_DEV void eval_uvw_cgrid(const KArgs& a, const CgLds& L, CCtxT<FT, CM>& c, double t, double z, double y, double x, bool pos_f32, double& u,
                           double& v, double& w, unsigned it, int klo, int sk = -1, double home_y = 0.0, double home_x = 0.0) {
    PK_CG_FMA
    const FastC& F = a.fastc;
    const bool scalar = WITH_SCALAR && sk >= 0;
    const int ks = scalar ? (sk & 1) : 0;
    u = v = w = 0.0;
    int ti = 0;
    double tau = 0.0;
    if (scalar ? F.kh_has_ti[ks] != 0 : F.has_ti != 0) {  // _search_time_index (index_search.py:65-91); (it, klo): pk_device.h, twe_note
        if (__builtin_expect(!(0 <= t) || !(t <= L.tlen), 0)) {
            c.state = PK_ERROROUTSIDETIMEINTERVAL;
            twe_note(a, it, klo);
            return;
        }
        if ((CM & CG_NO_TMEMO) != 0 || t != c.mt) {
            int idx;
            fast_search(L.time, F.nt, F.t0, F.t1, t, c.ht, idx, c.mtau);  // level times start at 0 (host check): idx == c.ht
            if constexpr ((CM & CG_NO_TMEMO) == 0) c.mt = t;
        }
        ti = c.ht;
        tau = c.mtau;
    }
    int zi = 0;
    double zeta = 0.0;
    if (F.has_z) {
        if (!(z == c.mz)) {
            fast_search(L.depth, F.gnz, F.z0, F.z1, z, c.hz, c.zi, c.mzeta);
            c.mz = z;
        }
        zi = c.zi;
        zeta = c.mzeta;
    }
    const bool lenT = tau > 0;
    // the query point on the unit sphere (make_qpoint / latlon_rad_to_xyz)
    double sl, cl, so, co;
    const bool guess_ok = c.gy >= 0 && c.gy < F.gny - 1 && c.gx >= 0 && c.gx < F.gnx - 1;
    if constexpr ((NEAR & 1) != 0) {  // (home_y, home_x): the particle's own position, whose sines / cosines the kernel left in the context
        const double dy = (y - home_y) * DEG2RAD, dx = (x - home_x) * DEG2RAD;
        if (__builtin_expect(fmax(fabs(dy), fabs(dx)) <= CG_NEAR_MAX, 1)) {
            // (AdvectionDiffusionM1: five of the seven samples of a step at the particle's own latitude, five at its own longitude --
            // the same for every lane of the wavefront)
            if (WITH_SCALAR && y == home_y) { sl = c.q_sl; cl = c.q_cl; } else sincos_near(dy, c.q_sl, c.q_cl, sl, cl);
            if (WITH_SCALAR && x == home_x) { so = c.q_so; co = c.q_co; } else sincos_near(dx, c.q_so, c.q_co, so, co);
        } else {  // (and NaN)
            sincos_geo(y * DEG2RAD, sl, cl);
            sincos_geo(x * DEG2RAD, so, co);
        }
    } else if (WITH_SCALAR) {
        if (y == home_y) { sl = c.q_sl; cl = c.q_cl; } else sincos_geo(y * DEG2RAD, sl, cl);
        if (x == home_x) { so = c.q_so; co = c.q_co; } else sincos_geo(x * DEG2RAD, so, co);
    } else {
        sincos_geo(y * DEG2RAD, sl, cl);
        sincos_geo(x * DEG2RAD, so, co);
    }
    const double qX = co * cl, qY = so * cl, qZ = sl;
    // _search_indices_curvilinear_2d with a guess (index_search.py:242-295); see curvilinear_search for the probing order
    int yi = GRID_SEARCH_ERROR, xi = GRID_SEARCH_ERROR;
    double xsi = -1.0, eta = -1.0;
    bool found = false;
    {
        if (__builtin_expect(guess_ok, 1)) {
            const int cell = c.gy * F.gnx + c.gx;
            if (c.rc_cell != cell) {  // (a scalar sample fetches the record only)
                if (scalar) cg_fetch_cell<FT, D3, false, CM>(F, L, c, cell, c.gy, c.gx, zi, ti, lenT);
                else cg_fetch_cell<FT, D3, true, CM>(F, L, c, cell, c.gy, c.gx, zi, ti, lenT);
            }
            double xs, et;
            if (cg_point_in_cell<CM>(F, L.rec, cell, qX, qY, qZ, xs, et)) {
                found = true;
                yi = c.gy;
                xi = c.gx;
                xsi = xs;
                eta = et;
            } else if (F.walk_ok) {
                {
                    // the particle left the guessed cell: the neighbour its barycentric coordinates point at (curvilinear_search)
                    int dj = et < 0 ? -1 : (et > 1 ? 1 : 0), di = xs < 0 ? -1 : (xs > 1 ? 1 : 0);
                    if constexpr (HOPS == 0) {  // one probe, but of the cell floor(xsi, eta) cells away (AdvectionRK45's long stages)
                        if (fabs(xs) < 64.0 && fabs(et) < 64.0) {
                            dj = et < 0 ? (int)floor(et) : (et > 1 ? (int)ceil(et) - 1 : 0);
                            di = xs < 0 ? (int)floor(xs) : (xs > 1 ? (int)ceil(xs) - 1 : 0);
                        }
                    }
                    const int nj = c.gy + dj, ni = c.gx + di;
                    if ((dj | di) != 0 && nj >= 0 && nj < F.gny - 1 && ni >= 0 && ni < F.gnx - 1) {
                        const int ncell = nj * F.gnx + ni;
                        const double boxd = scalar ? cg_fetch_cell<FT, D3, false, CM>(F, L, c, ncell, nj, ni, zi, ti, lenT) : cg_fetch_cell<FT, D3, true, CM>(F, L, c, ncell, nj, ni, zi, ti, lenT);
                        double xs2, et2;
                        const double m = 1e-9;
                        if (cg_point_in_cell<CM>(F, L.rec, ncell, qX, qY, qZ, xs2, et2) && xs2 > m && xs2 < 1 - m && et2 > m && et2 < 1 - m) {
                            if (box_lists(kgrid(a, F.grid), (unsigned long long)__double_as_longlong(boxd), qX, qY, qZ, true)) {
                                found = true;
                                yi = nj;
                                xi = ni;
                                xsi = (double)(float)xs2;  // rounded like a hash hit (spatialhash.py:505)
                                eta = (double)(float)et2;
                            }
                        }
                    }
                }
            }
        }
        if (__builtin_expect(!found, 0)) {
            // the faces of the query's hash cell in table order (SpatialHash.query, spatialhash.py:389-535): the general routine
            int wy, wx;
            double wxs, wet;
            curvilinear_search(kgrid(a, F.grid), y, x, false, 0, 0, wy, wx, wxs, wet);
            yi = wy;
            xi = wx;
            xsi = wxs;
            eta = wet;
        }
    }
    // ravel_index (basegrid.py:83-152): the low 32 bits of the int64 sum are the wrapped 32-bit sum
    c.ei = (int32_t)((uint32_t)xi * F.ex + (uint32_t)yi * F.ey + (uint32_t)zi * F.ez);
    if (__builtin_expect((xi | yi | zi) < 0, 0)) {
        int s = c.state;  // field.py:307-356
        if (xi == GRID_SEARCH_ERROR && s < PK_ERRORGRIDSEARCHING) s = PK_ERRORGRIDSEARCHING;
        if (zi == RIGHT_OUT_OF_BOUNDS && s < PK_ERROROUTOFBOUNDS) s = PK_ERROROUTOFBOUNDS;
        if (zi == LEFT_OUT_OF_BOUNDS && s < PK_ERRORTHROUGHSURFACE) s = PK_ERRORTHROUGHSURFACE;
        // field.py:359-378: a non-finite barycentric coordinate makes the (wrapped-around) gather NaN, then everything is zeroed
        const bool bad = !(isfinite(xsi) && isfinite(eta) && isfinite(zeta) && isfinite(tau));
        if (bad && s < PK_ERRORINTERPOLATION) s = PK_ERRORINTERPOLATION;
        c.state = s;
        // the next guess is what unravelling this `ei` gives (index_search.py:269-274)
        if (xi >= 0) { c.gy = yi; c.gx = xi; }
        else unravel_yx(kgrid(a, F.grid), (int64_t)c.ei, c.gy, c.gx);
        return;
    }
    c.gy = yi;
    c.gx = xi;
    if (scalar) {  // Field.eval: XLinear on the nodes of the cell, NaN -> ErrorInterpolation (field.py:373-378)
        const double val = cg_scalar_xlinear<FT>(F, ks, ti, tau, zi, zeta, yi, xi, xsi, eta);
        if (__builtin_expect(val != val, 0)) {
            if (c.state < PK_ERRORINTERPOLATION) c.state = PK_ERRORINTERPOLATION;
        }
        u = val;
        return;
    }
    const int cell = yi * F.gnx + xi;
    const bool need_rec = c.rc_cell != cell, need_f = !cg_fields_cached(c, cell, zi, ti, lenT);
    if (__builtin_expect(need_rec, 0)) {  // found by the table walk
        cg_fetch_cell<FT, D3, true, CM>(F, L, c, cell, yi, xi, zi, ti, lenT);
    } else if (need_f) {  // same cell, another depth or time level
        cg_refresh_fields<FT, D3, CM, true>(F, L, c, cell, zi, yi, xi, ti, lenT);
    }
    // ---- CGrid_Velocity.interp (_xinterpolators.py:193-332), float64 coordinates and barycentric arrays ----
    double px[4], py[4];
    if constexpr ((CM & CG_PXY_GLOBAL) != 0) {
        const double* g = L.ct2 + (int64_t)cell * CT2_STRIDE + 16;
        ldpair(g, px[0], px[1]);
        ldpair(g + 2, px[2], px[3]);
        ldpair(g + 4, py[0], py[1]);
        ldpair(g + 6, py[2], py[3]);
    } else if constexpr ((CM & CG_PXY_REGS) != 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) { px[k] = c.pxy[k]; py[k] = c.pxy[4 + k]; }
    } else {
        const double* rec = L.rec;  // (rows 16..23 of the record: slot rows 15..22, or the pairs 8..11 of CG_DMA)
        constexpr int r0 = (CM & CG_DMA) ? 16 : 15;
#pragma unroll
        for (int k = 0; k < 4; k++) { px[k] = cg_rec_row<CM>(rec, r0 + k); py[k] = cg_rec_row<CM>(rec, r0 + 4 + k); }
    }
    FT rawf[12];
    if constexpr ((CM & CG_FV_REGS) != 0) {
#pragma unroll
        for (int k = 0; k < 12; k++) rawf[k] = c.fvr[k];
    } else {
        const FT* fv = (const FT*)L.fv;
#pragma unroll
        for (int k = 0; k < 6; k++) rawf[k] = fv[k * FC_LANES];
        if (lenT) {
#pragma unroll
            for (int k = 0; k < 6; k++) rawf[6 + k] = fv[(6 + k) * FC_LANES];
        }
    }
    const double omx = 1 - xsi, ome = 1 - eta;
    // einsum("ij,ji->i", phi2D_lin(eta, xsi), py) at (0, xsi), (eta, 1), (1, xsi), (eta, 0): the products with an exact zero weight
    // add +-0 to a finite sum and are left out; 1 * w is w
    double c1, c2, c3, c4;
    if constexpr ((NEAR & 2) != 0) {
        // (kernels launched for a grid with FastC::near_edges) no cell of this grid spans 2^-8 rad of latitude: the edge points lie within
        // 2^-7 rad of the sample's latitude, whose (sl, cl) are at hand.  A compile-time choice: with both paths in one kernel the register
        // allocator spills 110 B per lane more in the RK45 kernel
        auto geodn = [&](double la1, double la2, double lo1, double lo2, double lat) {  // _geodetic_distance (utils/interpolation.py:178-185)
            const double dl = (lo2 - lo1) * F.deg2m, dla = (la2 - la1) * F.deg2m;
            const double aa_ = dl * cos_near((lat - y) * DEG2RAD, sl, cl);
            const double d2 = aa_ * aa_ + dla * dla;
            return PK_CG_LEAN ? (d2 > 0 ? sqrt_lean(d2) : d2) : sqrt(d2);  // (coincident corners: 0)
        };
        c1 = geodn(py[0], py[1], px[0], px[1], omx * py[0] + xsi * py[1]);
        c2 = geodn(py[1], py[2], px[1], px[2], ome * py[1] + eta * py[2]);
        c3 = geodn(py[2], py[3], px[2], px[3], xsi * py[2] + omx * py[3]);
        c4 = geodn(py[3], py[0], px[3], px[0], ome * py[0] + eta * py[3]);
    } else {
        auto geod = [&](double la1, double la2, double lo1, double lo2, double lat) {
            const double dl = (lo2 - lo1) * F.deg2m, dla = (la2 - la1) * F.deg2m;
            const double aa_ = dl * cos_lat(DEG2RAD * lat);
            const double d2 = aa_ * aa_ + dla * dla;
            return PK_CG_LEAN ? (d2 > 0 ? sqrt_lean(d2) : d2) : sqrt(d2);
        };
        c1 = geod(py[0], py[1], px[0], px[1], omx * py[0] + xsi * py[1]);
        c2 = geod(py[1], py[2], px[1], px[2], ome * py[1] + eta * py[2]);
        c3 = geod(py[2], py[3], px[2], px[3], xsi * py[2] + omx * py[3]);
        c4 = geod(py[3], py[0], px[3], px[0], ome * py[0] + eta * py[3]);
    }
    double raw[6];
#pragma unroll
    for (int k = 0; k < 6; k++) raw[k] = lenT ? (double)rawf[k] * (1 - tau) + (double)rawf[6 + k] * tau : (double)rawf[k];
    const double ua = raw[0], ub = raw[1], va = raw[2], vb = raw[3];
    const double U0 = ua * c4, U1 = ub * c2;
    const double Uvel = omx * U0 + xsi * U1;
    const double V0 = va * c1, V1 = vb * c3;
    const double Vvel = ome * V0 + eta * V1;
    // _compute_jacobian_determinant (utils/interpolation.py:188-198)
    const double dxs0 = eta - 1, dxs1 = ome, dxs2 = eta, dxs3 = -eta;
    const double det0 = xsi - 1, det1 = -xsi, det2 = xsi, det3 = omx;
    const double dxdxsi = ((dxs0 * px[0] + dxs1 * px[1]) + dxs2 * px[2]) + dxs3 * px[3];
    const double dxdeta = ((det0 * px[0] + det1 * px[1]) + det2 * px[2]) + det3 * px[3];
    const double dydxsi = ((dxs0 * py[0] + dxs1 * py[1]) + dxs2 * py[2]) + dxs3 * py[3];
    const double dydeta = ((det0 * py[0] + det1 * py[1]) + det2 * py[2]) + det3 * py[3];
    double jac = dxdxsi * dydeta - dxdeta * dydxsi;
    jac = jac * F.deg2m;  // spherical mesh
    const double A = -ome * Uvel - omx * Vvel;
    const double B = ome * Uvel - xsi * Vvel;
    const double C = eta * Uvel + xsi * Vvel;
    const double D = -eta * Uvel + omx * Vvel;
    double conv;  // :311-314 (both components divided by deg2m * cos(lat))
    if (PF && pos_f32) conv = (double)((float)F.deg2m * cosf((float)y * DEG2RADF));
    else conv = F.deg2m * cl;  // cl == cos_lat(y * DEG2RAD): same reduction, same kernels (pk_device.h)
    double uu = A * px[0] + B * px[1] + C * px[2] + D * px[3], vv = A * py[0] + B * py[1] + C * py[2] + D * py[3];
    const double jc = jac * conv;
    if (PK_CG_LEAN && __builtin_expect(fabs(jc) > 1e-280 && fabs(jc) < 1e280, 1)) {  // (x / jac) / conv as x * (1 / (jac * conv)): one reciprocal for both divisions of both components
        const double r = rcp_lean(jc);
        uu = uu * r;
        vv = vv * r;
    } else {  // a degenerate cell (jac == 0: the reference's inf / NaN), non-finite values
        const Recip rjac = make_recip(jac);
        uu = div_shared(uu, rjac);
        vv = div_shared(vv, rjac);
        const Recip rconv = make_recip(conv);
        uu = div_shared(uu, rconv);
        vv = div_shared(vv, rconv);
    }
    double ww = 0.0;
    if (D3) ww = raw[4] * (1 - zeta) + raw[5] * zeta;  // :316-328
    if (__builtin_expect(uu != uu || vv != vv || ww != ww, 0)) {  // field.py:373-378
        if (c.state < PK_ERRORINTERPOLATION) c.state = PK_ERRORINTERPOLATION;
    }
    u = uu;
    v = vv;
    w = ww;
}

}  // namespace pk
