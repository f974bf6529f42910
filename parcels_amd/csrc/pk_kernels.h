// pk_kernels.h -- the fused advection kernel: Kernel.execute (kernel.py:174-247) for one particle per lane.
//
// A launch advances every live particle from its own t to `endtime`: dt clipping, the kernel list, the
// RK45 Repeat loop, the position update, EndofLoop marking and the status-code state machine all run in
// registers; particle state is read once and written once per launch (not once per step as in the NumPy
// reference), so HBM traffic is the field gather only.
//
// Every built-in kernel is written as a small stage machine: prepare(stage) names the next sample point,
// the step loop performs the field evaluation at ONE call site, consume(stage) folds the result into the
// kernel's registers.  That keeps the code (and its instruction-cache footprint) at one copy of the
// search+gather+interpolate sequence per kernel instead of one per Runge-Kutta stage, and lets the same
// source serve the specialised programs (KID fixed at compile time: the switch folds away) and the generic
// kernel-list program (KID = -1).
#pragma once
#include "pk_device.h"
#include "pk_fast_agrid.h"
#include "pk_fast_cgrid.h"

namespace pk {

static_assert(sizeof(KArgs) <= 4096, "kernel arguments must fit the 4 KiB kernarg segment");

// programs: a single built-in kernel fixed at compile time, or the generic kernel-list interpreter
enum Program { PROG_RK4 = 0, PROG_RK4_3D = 1, PROG_GENERIC = 2, PROG_RK45 = 3, PROG_M1 = 4, PROG_TYPED = 5 };  // PROG_TYPED: the kernel-list
// interpreter with NumPy's float32 dtype propagation (fieldsets with float32 coordinate arrays; pk_device.h: TYPED)

struct PState {
    double t, z, y, x, dz, dy, dx, dt, next_dt;
    int64_t id;
};

PK_DEV double ldp(const void* col, int64_t i, bool pf) { return pf ? (double)((const float*)col)[i] : ((const double*)col)[i]; }
PK_DEV void stp(void* col, int64_t i, double v, bool pf) {
    if (pf) ((float*)col)[i] = (float)v; else ((double*)col)[i] = v;
}
// storage-dtype arithmetic of the particle columns (particlesetview.py:202-205): f32 + f32 is an f32 add
PK_DEV double padd(bool pf, double a, double b) { return pf ? (double)((float)a + (float)b) : a + b; }
PK_DEV double psub(bool pf, double a, double b) { return pf ? (double)((float)a - (float)b) : a - b; }
PK_DEV double pstore(bool pf, double v) { return pf ? (double)(float)v : v; }

// what the current stage wants sampled
enum { RQ_UV = 0, RQ_UVW = 1, RQ_SCALAR = 2 };
struct Request {
    int kind, fidx;
    double t, z, y, x;
    bool f32;    // sample point (y, x) comes straight from float32 particle storage
    bool zf32;   // its z does: every sample at `particles.z` (all stages of the 2-D kernels), whatever y and x are (PCtx::zpos_f32)
    bool reuse;  // scalar sample at the point of this kernel's velocity sample: the grid position may be re-used (SearchMemo)
};

// kernel-local registers (all indices are compile-time constants -> scalar-replaced into VGPRs)
struct KLocal {
    double r[14];
    bool u1f, v1f;  // AdvectionRK45: u1 / v1 are float32 ARRAYS in the reference (PCtx::u32 / v32 of the stage-0 sample)
#ifdef PK_USER_KERNELS
    PkUserLocals ul;  // locals of the user kernel being run (generated struct, parcels_amd/jit.py)
#endif
};

// meters_to_degrees_zonal / _meridional (_advectiondiffusion.py:11-18); `particles.y * np.pi / 180` is float32
// arithmetic for the default Particle
PK_DEV double m2_to_deg2_zonal(bool pf, double v, double lat, double deg2m) {
    if (pf) {
        float a_ = (float)deg2m * cosf((float)lat * 3.14159265358979323846f / 180.0f);
        return v / (double)(a_ * a_);
    }
    double a_ = deg2m * cos_lat(lat * 3.14159265358979323846 / 180);  // (pk_device.h: < 1 ulp, like the reference's libm)
    return v / (a_ * a_);
}
PK_DEV double m2_to_deg2_merid(double v, double deg2m) { return v / (deg2m * deg2m); }

// Fehlberg tableau of AdvectionRK45 (_advection.py:96-106)
namespace rk45c {
constexpr double c0 = 1.0 / 4.0, c1 = 3.0 / 8.0, c2 = 12.0 / 13.0, c3 = 1.0, c4 = 1.0 / 2.0;
constexpr double A00 = 1.0 / 4.0;
constexpr double A10 = 3.0 / 32.0, A11 = 9.0 / 32.0;
constexpr double A20 = 1932.0 / 2197.0, A21 = -7200.0 / 2197.0, A22 = 7296.0 / 2197.0;
constexpr double A30 = 439.0 / 216.0, A31 = -8.0, A32 = 3680.0 / 513.0, A33 = -845.0 / 4104.0;
constexpr double A40 = -8.0 / 27.0, A41 = 2.0, A42 = -3544.0 / 2565.0, A43 = 1859.0 / 4104.0, A44 = -11.0 / 40.0;
constexpr double b40 = 25.0 / 216.0, b41 = 0.0, b42 = 1408.0 / 2565.0, b43 = 2197.0 / 4104.0, b44 = -1.0 / 5.0;
constexpr double b50 = 16.0 / 135.0, b51 = 0.0, b52 = 6656.0 / 12825.0, b53 = 28561.0 / 56430.0, b54 = -9.0 / 50.0,
                 b55 = 2.0 / 55.0;
}  // namespace rk45c

#ifdef PK_USER_KERNELS
// User-written kernels translated from Python (parcels_amd/jit.py), defined by the generated translation unit after this header: stage 0
// runs the statements up to the first field sample (and names it in rq), stage k those after the k-th sample (its values are in
// L.r[3..5], where consume() leaves the latest sample), the last stage returns true.
PK_DEV bool user_prepare(const KArgs& a, int uk, int stage, int kslot, PCtx& c, PState& p, KLocal& L, Request& rq);
#endif
// prepare(): returns true when the kernel is finished (after writing its result into p / c), otherwise fills rq.
PK_DEV bool prepare(const KArgs& a, int kid, int stage, int kslot, PCtx& c, PState& p, KLocal& L, Request& rq) {
    const pk_exec_params& prm = a.prm;
    const bool pf = c.pf;
    rq.kind = RQ_UV;
    rq.fidx = 0;
    rq.f32 = false;
    rq.zf32 = pf;  // (rq.z = p.z below: a kernel that samples at another depth says so)
    rq.reuse = false;
    rq.t = p.t; rq.z = p.z; rq.y = p.y; rq.x = p.x;
    switch (kid) {
        case PK_KERNEL_ADVECTION_RK4:
        case PK_KERNEL_ADVECTION_RK4_3D: {  // _advection.py:42-75; r[0..2] running sum, r[3..5] last u,v,w
            const bool d3 = kid == PK_KERNEL_ADVECTION_RK4_3D;
            rq.kind = d3 ? RQ_UVW : RQ_UV;
            if (stage == 0) { rq.f32 = pf; return false; }
            if (stage == 4) {
                p.dx = pstore(pf, p.dx + L.r[0] / 6.0 * p.dt);
                p.dy = pstore(pf, p.dy + L.r[1] / 6.0 * p.dt);
                if (d3) p.dz = pstore(pf, p.dz + L.r[2] / 6 * p.dt);
                return true;
            }
            const double cdt = stage == 3 ? 1.0 : 0.5;  // u*1.0 == u and 1.0*dt == dt exactly
            rq.x = p.x + L.r[3] * cdt * p.dt;
            rq.y = p.y + L.r[4] * cdt * p.dt;
            if (d3) { rq.z = p.z + L.r[5] * cdt * p.dt; rq.zf32 = false; }
            rq.t = p.t + cdt * p.dt;
            return false;
        }
        case PK_KERNEL_ADVECTION_RK2:
        case PK_KERNEL_ADVECTION_RK2_3D: {  // _advection.py:21-39
            const bool d3 = kid == PK_KERNEL_ADVECTION_RK2_3D;
            rq.kind = d3 ? RQ_UVW : RQ_UV;
            if (stage == 0) { rq.f32 = pf; return false; }
            if (stage == 2) {
                p.dx = pstore(pf, p.dx + L.r[3] * p.dt);
                p.dy = pstore(pf, p.dy + L.r[4] * p.dt);
                if (d3) p.dz = pstore(pf, p.dz + L.r[5] * p.dt);
                return true;
            }
            rq.x = p.x + L.r[3] * 0.5 * p.dt;
            rq.y = p.y + L.r[4] * 0.5 * p.dt;
            if (d3) { rq.z = p.z + L.r[5] * 0.5 * p.dt; rq.zf32 = false; }
            rq.t = p.t + 0.5 * p.dt;
            return false;
        }
        case PK_KERNEL_ADVECTION_EE: {  // _advection.py:78-82
            if (stage == 0) { rq.f32 = pf; return false; }
            p.dx = pstore(pf, p.dx + L.r[3] * p.dt);
            p.dy = pstore(pf, p.dy + L.r[4] * p.dt);
            return true;
        }
        case PK_KERNEL_ADVECTION_RK45: {  // _advection.py:85-155; r[0..5] = u1..u6, r[6..11] = v1..v6
            using namespace rk45c;
            const double dt = p.dt;
            const double *u = &L.r[0], *v = &L.r[6];
            // u1 / v1 may be float32 ARRAYS in the reference (float32 data on float32 coordinates sampled at float32 positions, or
            // the unguessed first evaluation on a curvilinear A-grid): their product with a Python float is a float32 product
            // (NEP 50).  Every later sample point is float64, so only the stage-1 terms are affected.
            auto U1 = [&](double k) { return L.u1f ? (double)((float)u[0] * (float)k) : u[0] * k; };
            auto V1 = [&](double k) { return L.v1f ? (double)((float)v[0] * (float)k) : v[0] * k; };
            switch (stage) {
                case 0: rq.f32 = pf; return false;
                case 1:
                    rq.x = p.x + U1(A00) * dt; rq.y = p.y + V1(A00) * dt; rq.t = p.t + c0 * dt;
                    return false;
                case 2:
                    rq.x = p.x + (U1(A10) + u[1] * A11) * dt; rq.y = p.y + (V1(A10) + v[1] * A11) * dt;
                    rq.t = p.t + c1 * dt;
                    return false;
                case 3:
                    rq.x = p.x + (U1(A20) + u[1] * A21 + u[2] * A22) * dt;
                    rq.y = p.y + (V1(A20) + v[1] * A21 + v[2] * A22) * dt;
                    rq.t = p.t + c2 * dt;
                    return false;
                case 4:
                    rq.x = p.x + (U1(A30) + u[1] * A31 + u[2] * A32 + u[3] * A33) * dt;
                    rq.y = p.y + (V1(A30) + v[1] * A31 + v[2] * A32 + v[3] * A33) * dt;
                    rq.t = p.t + c3 * dt;
                    return false;
                case 5:
                    rq.x = p.x + (U1(A40) + u[1] * A41 + u[2] * A42 + u[3] * A43 + u[4] * A44) * dt;
                    rq.y = p.y + (V1(A40) + v[1] * A41 + v[2] * A42 + v[3] * A43 + v[4] * A44) * dt;
                    rq.t = p.t + c4 * dt;
                    return false;
                default: break;
            }
            const double sign_dt = (dt > 0) ? 1.0 : ((dt < 0) ? -1.0 : dt);  // np.sign
            const double x_4th = (U1(b40) + u[1] * b41 + u[2] * b42 + u[3] * b43 + u[4] * b44) * dt;
            const double y_4th = (V1(b40) + v[1] * b41 + v[2] * b42 + v[3] * b43 + v[4] * b44) * dt;
            const double x_5th = (U1(b50) + u[1] * b51 + u[2] * b52 + u[3] * b53 + u[4] * b54 + u[5] * b55) * dt;
            const double y_5th = (V1(b50) + v[1] * b51 + v[2] * b52 + v[3] * b53 + v[4] * b54 + v[5] * b55) * dt;
            const double ex = x_5th - x_4th, ey = y_5th - y_4th;
            const double kappa = sqrt(ex * ex + ey * ey);
            const bool good = (kappa <= prm.rk45_tol) || (fabs(dt) <= fabs(prm.rk45_min_dt));
            p.dx = pstore(pf, p.dx + (good ? x_5th : 0.0));
            p.dy = pstore(pf, p.dy + (good ? y_5th : 0.0));
            const bool increase = good && (kappa <= prm.rk45_tol / 10) && (fabs(dt * 2) <= fabs(prm.rk45_max_dt));
            double next_dt = increase ? dt * 2 : dt;
            if (fabs(next_dt) > fabs(prm.rk45_max_dt)) next_dt = prm.rk45_max_dt * sign_dt;
            p.next_dt = prm.next_dt_f32 ? (double)(float)next_dt : next_dt;  // assignment into an f32 column
            if (good) c.state = PK_EVALUATE;  // :146 overwrites any sampling error code
            double ndt = good ? dt : dt / 2;
            if (fabs(ndt) < fabs(prm.rk45_min_dt)) ndt = prm.rk45_min_dt * sign_dt;
            p.dt = ndt;
            if (!good) c.state = PK_REPEAT;
            return true;
        }
        case PK_KERNEL_ADVECTIONDIFFUSION_M1:
        case PK_KERNEL_ADVECTIONDIFFUSION_EM: {
            // _advectiondiffusion.py:21-117.  evaluation order M1: Kxp1 Kxm1 UV khz Kyp1 Kym1 khm ; EM: UV Kxp1 Kxm1 khz ...
            // r[0] Kxp1, r[1] Kxm1, r[2] u, r[3] v, r[4] khz, r[5] Kyp1, r[6] Kym1, r[7] khm
            const bool em = kid == PK_KERNEL_ADVECTIONDIFFUSION_EM;
            const double dres = prm.dres;
            int what = stage;  // M1 order
            if (em) what = stage == 0 ? 2 : (stage <= 2 ? stage - 1 : stage);
            rq.f32 = pf;
            switch (stage < 7 ? what : 7) {
                case 0: rq.kind = RQ_SCALAR; rq.fidx = prm.fKh_zonal; rq.x = padd(pf, p.x, dres); return false;
                case 1: rq.kind = RQ_SCALAR; rq.fidx = prm.fKh_zonal; rq.x = psub(pf, p.x, dres); return false;
                case 2: rq.kind = RQ_UV; return false;
                case 3: rq.kind = RQ_SCALAR; rq.fidx = prm.fKh_zonal; rq.reuse = true; return false;
                case 4: rq.kind = RQ_SCALAR; rq.fidx = prm.fKh_meridional; rq.y = padd(pf, p.y, dres); return false;
                case 5: rq.kind = RQ_SCALAR; rq.fidx = prm.fKh_meridional; rq.y = psub(pf, p.y, dres); return false;
                case 6: rq.kind = RQ_SCALAR; rq.fidx = prm.fKh_meridional; rq.reuse = true; return false;
                default: break;
            }
            const DGrid& gz = kgrid(a, kfield(a, prm.fKh_zonal).grid);
            const DGrid& gm = kgrid(a, kfield(a, prm.fKh_meridional).grid);
            double Kxp1 = L.r[0], Kxm1 = L.r[1], khz = L.r[4], Kyp1 = L.r[5], Kym1 = L.r[6], khm = L.r[7];
            const double u = L.r[2], v = L.r[3];
            if (gz.spherical) {
                Kxp1 = m2_to_deg2_zonal(pf, Kxp1, p.y, gz.deg2m);
                Kxm1 = m2_to_deg2_zonal(pf, Kxm1, p.y, gz.deg2m);
                khz = m2_to_deg2_zonal(pf, khz, p.y, gz.deg2m);
            }
            if (gm.spherical) {
                Kyp1 = m2_to_deg2_merid(Kyp1, gm.deg2m);
                Kym1 = m2_to_deg2_merid(Kym1, gm.deg2m);
                khm = m2_to_deg2_merid(khm, gm.deg2m);
            }
            const double dKdx = (Kxp1 - Kxm1) / (2 * dres), dKdy = (Kyp1 - Kym1) / (2 * dres);
            const double bx = sqrt(2 * khz), by = sqrt(2 * khm);
            double z0, z1;
            normal_pair(prm.seed, kslot, p.id, p.t, z0, z1);
            const double s = sqrt(fabs(p.dt));
            const double dWx = s * z0, dWy = s * z1;
            if (!em) {  // :66-67
                p.dx = pstore(pf, p.dx + (u * p.dt + 0.5 * dKdx * (dWx * dWx + p.dt) + bx * dWx));
                p.dy = pstore(pf, p.dy + (v * p.dt + 0.5 * dKdy * (dWy * dWy + p.dt) + by * dWy));
            } else {  // :116-117
                const double ax = u + dKdx, ay = v + dKdy;
                p.dx = pstore(pf, p.dx + (ax * p.dt + bx * dWx));
                p.dy = pstore(pf, p.dy + (ay * p.dt + by * dWy));
            }
            return true;
        }
        case PK_KERNEL_DIFFUSION_UNIFORM_KH: {  // _advectiondiffusion.py:120-153; r[0] kh_zonal, r[1] kh_meridional
            rq.f32 = pf;
            if (stage == 0) { rq.kind = RQ_SCALAR; rq.fidx = prm.fKh_zonal; return false; }
            if (stage == 1) { rq.kind = RQ_SCALAR; rq.fidx = prm.fKh_meridional; return false; }
            const DGrid& gz = kgrid(a, kfield(a, prm.fKh_zonal).grid);
            const DGrid& gm = kgrid(a, kfield(a, prm.fKh_meridional).grid);
            double khz = L.r[0], khm = L.r[1];
            if (gz.spherical) {
                khz = m2_to_deg2_zonal(pf, khz, p.y, gz.deg2m);
                khm = m2_to_deg2_merid(khm, gm.deg2m);
            }
            double z0, z1;
            normal_pair(prm.seed, kslot, p.id, p.t, z0, z1);
            const double s = sqrt(fabs(p.dt));
            p.dx = pstore(pf, p.dx + sqrt(2 * khz) * (s * z0));
            p.dy = pstore(pf, p.dy + sqrt(2 * khm) * (s * z1));
            return true;
        }
        case PK_KERNEL_SAMPLE_FIELD: {  // particles.<var> = fieldset.<F>[particles] (field.py:187-195: eval at the particle's t, z, y, x)
            const int sf = prm.sample_field[kslot];
            if (stage == 0) {
                rq.kind = sf >= 0 ? RQ_SCALAR : (sf == PK_SAMPLE_UVW ? RQ_UVW : RQ_UV);
                rq.fidx = sf >= 0 ? sf : 0;
                rq.f32 = pf;
                return false;
            }
            // assignment into the Variable's dtype (particlesetview.py:202-205); a vector sample fills up to three Variables
            const int nv = sf >= 0 ? 1 : (sf == PK_SAMPLE_UVW ? 3 : 2);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int v = (prm.sample_var[kslot] >> (8 * j)) & 0xFF;
                if (j >= nv || v == PK_SAMPLE_DISCARD) continue;
                if (a.p.extra_f32[v]) ((float*)a.p.extra[v])[c.row] = (float)L.r[j];
                else ((double*)a.p.extra[v])[c.row] = L.r[j];
            }
            return true;
        }
        case PK_KERNEL_DO_NOTHING:  // tests/common_kernels.py:8-9
            return true;
        case PK_KERNEL_MOVE_EAST:  // tests/common_kernels.py:16-17: `particles.dx += 0.1` (an in-place add in the storage dtype)
            p.dx = padd(pf, p.dx, 0.1);
            return true;
        case PK_KERNEL_MOVE_NORTH:  // tests/common_kernels.py:20-21
            p.dy = padd(pf, p.dy, 0.1);
            return true;
        case PK_KERNEL_DELETE_ON_ERROR:  // tests/common_kernels.py:12-13
            if (c.state >= PK_ERROR) c.state = PK_DELETE;
            return true;
        case PK_KERNEL_DELETE_OUT_OF_BOUNDS:  // tests/test_advection.py:157-161
            if (c.state == PK_ERROROUTOFBOUNDS || c.state == PK_ERRORTHROUGHSURFACE) c.state = PK_DELETE;
            return true;
        case PK_KERNEL_SUBMERGE_THROUGH_SURFACE:  // tests/test_advection.py:163-174
            if (stage == 0) {
                if (c.state != PK_ERRORTHROUGHSURFACE) return true;
                rq.f32 = pf;
                return false;
            }
            p.dx = pstore(pf, L.r[3] * p.dt);
            p.dy = pstore(pf, L.r[4] * p.dt);
            p.dz = 0.0;
            p.z = 0.0;
            c.state = PK_EVALUATE;
            return true;
        default:
#ifdef PK_USER_KERNELS
            if (kid >= PK_KERNEL_USER0 && kid < PK_KERNEL_USER0 + PK_MAX_USER_KERNELS) return user_prepare(a, kid - PK_KERNEL_USER0, stage, kslot, c, p, L, rq);
#endif
            c.state = PK_ERROR;
            return true;
    }
}

// consume(): fold the sampled (u, v, w) of `stage` into the kernel's registers
PK_DEV void consume(int kid, int stage, const PCtx& c, KLocal& L, double u, double v, double w) {
    switch (kid) {
        case PK_KERNEL_ADVECTION_RK4:
        case PK_KERNEL_ADVECTION_RK4_3D:
            // (u1 + 2*u2 + 2*u3 + u4), summed left to right
            if (stage == 0) { L.r[0] = u; L.r[1] = v; L.r[2] = w; }
            else if (stage == 3) { L.r[0] = L.r[0] + u; L.r[1] = L.r[1] + v; L.r[2] = L.r[2] + w; }
            else { L.r[0] = L.r[0] + 2 * u; L.r[1] = L.r[1] + 2 * v; L.r[2] = L.r[2] + 2 * w; }
            L.r[3] = u; L.r[4] = v; L.r[5] = w;
            break;
        case PK_KERNEL_ADVECTION_RK45:
            switch (stage) {
                case 0: L.r[0] = u; L.r[6] = v; L.u1f = c.u32; L.v1f = c.v32; break;
                case 1: L.r[1] = u; L.r[7] = v; break;
                case 2: L.r[2] = u; L.r[8] = v; break;
                case 3: L.r[3] = u; L.r[9] = v; break;
                case 4: L.r[4] = u; L.r[10] = v; break;
                default: L.r[5] = u; L.r[11] = v; break;
            }
            break;
        case PK_KERNEL_ADVECTIONDIFFUSION_M1:
            switch (stage) {
                case 0: L.r[0] = u; break;
                case 1: L.r[1] = u; break;
                case 2: L.r[2] = u; L.r[3] = v; break;
                case 3: L.r[4] = u; break;
                case 4: L.r[5] = u; break;
                case 5: L.r[6] = u; break;
                default: L.r[7] = u; break;
            }
            break;
        case PK_KERNEL_ADVECTIONDIFFUSION_EM:
            switch (stage) {
                case 0: L.r[2] = u; L.r[3] = v; break;
                case 1: L.r[0] = u; break;
                case 2: L.r[1] = u; break;
                case 3: L.r[4] = u; break;
                case 4: L.r[5] = u; break;
                case 5: L.r[6] = u; break;
                default: L.r[7] = u; break;
            }
            break;
        case PK_KERNEL_DIFFUSION_UNIFORM_KH:
            if (stage == 0) L.r[0] = u; else L.r[1] = u;
            break;
        case PK_KERNEL_SAMPLE_FIELD:
            L.r[0] = u; L.r[1] = v; L.r[2] = w;
            break;
        default:  // RK2, RK2_3D, EE, Submerge: only the latest sample matters
            L.r[3] = u; L.r[4] = v; L.r[5] = w;
            break;
    }
}

// Lanes that do not step in a launch (their state was not Evaluate when a paused call is continued; not selected by the mask of a
// body_only launch) keep their row: the host copies the input columns to the output columns before such a launch (pk_api.hip:
// pk_execute_begin, device-to-device), so the kernels never branch into copy code -- a copy block in the entry branch of advect_kernel
// made the compiler keep a private copy of the 3.9 KB kernel-argument struct (4 KB of scratch per lane, M1 82 -> 203 ms).
// kernel.py:236-245: the first iteration (1-based, counted per particle since the Kernel.execute call began) that left a particle in
// an error state or StopAllExecution -- the reference raises / returns after THAT iteration of its batch loop
PK_DEV void note_error_iteration(const KArgs& a, int state, unsigned it) {
    if (state >= PK_ERROR || state == PK_STOPALLEXECUTION) atomicMin(&a.counters->err_iter, it);
}

PK_DEV unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// blockIdx -> logical tile: the dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md); give every XCD a
// contiguous range of (cell-sorted) particles so that its private L2 sees one compact region of the fields.
PK_DEV unsigned xcd_swizzle(unsigned bid, unsigned nb) {
    const unsigned xcd = bid & 7u, j = bid >> 3;
    const unsigned per = nb >> 3, rem = nb & 7u;
    return xcd * per + (xcd < rem ? xcd : rem) + j;
}

// KID: compile-time kernel id of a single-kernel program, or -1 for the kernel-list interpreter
#ifndef PK_MIN_WAVES
#define PK_MIN_WAVES 1
#endif
// curvilinear search / C-grid interpolation need more registers: fewer waves, fewer spills
#ifndef PK_MIN_WAVES_HEAVY
#define PK_MIN_WAVES_HEAVY 2
#endif
// PFM: particle storage dtype -- 0 float64, 1 float32, -1 decided at run time (P.spatial_f32).
template <class FT, int KIND, int INTERP, int KID, bool LDS, bool TYPED, int PFM = -1>
__global__ void __launch_bounds__(wg_size(KIND, LDS), (KIND == 1 || INTERP != 0) ? PK_MIN_WAVES_HEAVY : PK_MIN_WAVES) advect_kernel(const KArgs a) {
    constexpr int WG = wg_size(KIND, LDS);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DField& mf = kfield(a, a.main_field);
    const DGrid& mg = kgrid(a, a.main_grid);
    Coords mc;
    if (LDS) {
        // stage the 1-D coordinate vectors of the main grid once per workgroup (coalesced), search them from LDS
        double* s_time = smem + a.lds_time;
        double* s_depth = smem + a.lds_depth;
        double* s_lat = smem + a.lds_lat;
        double* s_lon = smem + a.lds_lon;
        const int nt = mf.has_time_interval ? mf.nt : 0;
        for (int k = threadIdx.x; k < nt; k += WG) s_time[k] = mf.time[k];
        const int nz = mg.has_z ? mg.nz : 0;
        for (int k = threadIdx.x; k < nz; k += WG) s_depth[k] = mg.depth[k];
        if (KIND == 0) {
            for (int k = threadIdx.x; k < mg.ny; k += WG) s_lat[k] = mg.lat[k];
            for (int k = threadIdx.x; k < mg.nx; k += WG) s_lon[k] = mg.lon[k];
        }
        __syncthreads();
        mc.time = s_time;
        mc.depth = s_depth;
        mc.lat = (KIND == 0) ? s_lat : mg.lat;
        mc.lon = (KIND == 0) ? s_lon : mg.lon;
    } else {
        mc.time = mf.time;
        mc.depth = mg.depth;
        mc.lat = mg.lat;
        mc.lon = mg.lon;
    }
    mc.cc.nodes = nullptr;
    mc.cc.key = nullptr;
    mc.cc.fvals = nullptr;
    if (LDS && KIND == 1 && a.lds_cc_nodes >= 0) {  // lane-private slots: no barrier needed
        mc.cc.nodes = smem + a.lds_cc_nodes + threadIdx.x;
        mc.cc.key = (int*)(smem + a.lds_cc_keys) + threadIdx.x;
        if (INTERP == 1 && a.lds_cc_fvals >= 0) mc.cc.fvals = (void*)((FT*)(smem + a.lds_cc_fvals) + threadIdx.x);
        mc.cc.key[0] = -1;
        mc.cc.key[CC_LANES] = -1;
    }
    mc.t0 = mf.tfirst;
    mc.t1 = mf.tlast;
    mc.z0 = mg.zfirst; mc.z1 = mg.zlast;
    mc.y0 = mg.yfirst; mc.y1 = mg.ylast;
    mc.x0 = mg.xfirst; mc.x1 = mg.xlast;

    const int64_t i = (int64_t)xcd_swizzle(blockIdx.x, gridDim.x) * WG + threadIdx.x;
    unsigned long long steps = 0, attempts = 0, paused = 0;
    if (i < a.p.n) {
        const DParticles& P = a.p;
        const DPOut& O = a.po;
        const pk_exec_params& prm = a.prm;
        PCtx c;
        const bool pf = PFM < 0 ? (P.spatial_f32 != 0) : (PFM == 1);
        c.pf = pf;
        c.row = i;
        // body_only: the caller runs the loop of kernel.py:190-245 itself (Python kernels between device kernels) and asks for ONE pass of
        // the kernel list over the particles the reference would evaluate now (kernel.py:193-195); states are the caller's
        const bool body = prm.body_only != 0;
        c.state = (prm.reset_state && !body) ? PK_EVALUATE : P.state[i];  // kernel.py:188
        const bool run = body ? P.iter[i] != 0  // the caller's `evaluate_particles` mask of this iteration (pk_particles_set_mask): every kernel
                                                 // of one iteration sees the same particles, whatever an earlier kernel did to their state
                              : c.state == PK_EVALUATE;
        if (run) {
            unsigned it = (prm.reset_state || body) ? 0u : (unsigned)P.iter[i];
            c.it = 0u;
            c.klo = 0;
            c.hz = c.hy = c.hx = c.ht = 0;
            c.hyx_valid = false;
            c.first_eval = prm.reset_state ? 0xFu : 0u;
            c.u32 = c.v32 = false;
            PState p;
            p.t = P.t[i];
            p.z = ldp(P.z, i, pf);
            p.y = ldp(P.y, i, pf);
            p.x = ldp(P.x, i, pf);
            p.dz = ldp(P.dz, i, pf);
            p.dy = ldp(P.dy, i, pf);
            p.dx = ldp(P.dx, i, pf);
            p.dt = P.dt[i];
            p.next_dt = P.next_dt ? P.next_dt[i] : 0.0;
            p.id = P.particle_id[i];
            const int ng = P.ngrids;
            c.ei0 = P.ei[i * ng];
            c.ei1 = ng > 1 ? P.ei[i * ng + 1] : 0;
            c.ei2 = ng > 2 ? P.ei[i * ng + 2] : 0;
            c.ei3 = ng > 3 ? P.ei[i * ng + 3] : 0;
            const double endtime = prm.endtime;
            const int sign = prm.dt0 > 0 ? 1 : -1;  // kernel.py:186
            const bool windowed = a.win_lo > -INFINITY || a.win_hi < INFINITY;  // some field streams through a ring of levels
            const int nk = KID >= 0 ? 1 : prm.nk;
            bool once = body;
            while (once || (!body && (c.state == PK_EVALUATE || c.state == PK_REPEAT))) {  // :190
                once = false;
                const double tte = sign * (endtime - p.t);
                if (!body && !(tte >= 0)) break;  // :193-197 (state is Evaluate here)
                if (prm.max_iters > 0 && it >= (unsigned)prm.max_iters) break;  // pk_execute_rerun: stop where the reference raised
                double dtc;
                if (sign == 1) dtc = fmax(fmin(p.dt, tte), 0.0);  // :200-203
                else dtc = fmin(fmax(p.dt, -tte), 0.0);
                if (body) dtc = p.dt;  // the caller clipped dt (for ALL particles, like :199-203)
                if (windowed && !body) {
                    // field-slab streaming: only step while [t, t+dt] lies inside the resident time window;
                    // otherwise leave the particle untouched (state Evaluate) for the next launch
                    const double t1 = p.t + dtc;
                    const double lo = fmin(p.t, t1), hi = fmax(p.t, t1);
                    if (lo < a.win_lo || hi > a.win_hi) { paused = 1; break; }
                }
                it++;
                c.it = body ? 0u : it;  // (a body_only launch is one iteration of the CALLER's loop: it handles the call-wide time error itself)
                p.dt = dtc;
                for (int k = 0; k < nk; k++) {  // :206-216
                    const int kid = KID >= 0 ? KID : prm.kernels[k];
                    c.klo = k * 1000;  // key of the kernel's first sample; the samples of a Repeat re-run count on (pk_device.h: twe_listed)
                    do {
                        KLocal L;
                        L.u1f = L.v1f = false;
                        Request rq;
                        // the kernels that sample scalars at the point of their velocity sample: AdvectionDiffusionM1 has its own
                        // program (KID), AdvectionDiffusionEM runs in the kernel-list interpreter (KID < 0)
                        constexpr bool MEMO = KID == PK_KERNEL_ADVECTIONDIFFUSION_M1 || KID < 0;
                        SearchMemo memo;
                        memo.grid = -1;
                        attempts++;
                        for (int stage = 0; !prepare(a, kid, stage, k, c, p, L, rq); stage++) {
                            double u, v = 0.0, w = 0.0;
                            c.zpos_f32 = rq.zf32;
                            if (rq.kind == RQ_SCALAR) {
                                u = eval_scalar<FT, TYPED>(a, mc, c, rq.fidx, rq.t, rq.z, rq.y, rq.x, rq.f32, (MEMO && rq.reuse) ? &memo : nullptr);
                            } else {
                                const bool keep = MEMO && (kid == PK_KERNEL_ADVECTIONDIFFUSION_M1 || kid == PK_KERNEL_ADVECTIONDIFFUSION_EM);
                                eval_uvw<FT, KIND, INTERP, TYPED>(a, mc, c, rq.kind == RQ_UVW, rq.t, rq.z, rq.y, rq.x, rq.f32, u, v, w, keep ? &memo : nullptr);
                            }
                            consume(kid, stage, c, L, u, v, w);
                        }
                    } while (c.state == PK_REPEAT);
                }
                if (body) break;  // position update, dt reset and EndofLoop belong to the caller's loop
                if (KID >= 0) {
                    // single-kernel programs may be followed by the sampling-free recovery kernels (Delete*)
                    for (int k = 1; k < prm.nk; k++) {
                        const int kid = prm.kernels[k];
                        attempts++;
                        if (kid == PK_KERNEL_DELETE_ON_ERROR) {
                            if (c.state >= PK_ERROR) c.state = PK_DELETE;
                        } else if (c.state == PK_ERROROUTOFBOUNDS || c.state == PK_ERRORTHROUGHSURFACE) {
                            c.state = PK_DELETE;  // PK_KERNEL_DELETE_OUT_OF_BOUNDS
                        }
                    }
                }
                if (c.state == PK_EVALUATE || c.state == PK_SUCCESS) {  // :219-222 -> _position_update :108-120
                    if (tte > 0 && p.t + p.dt == p.t) {
                        // The step would not advance t before endtime (dt == 0 -- e.g. RK45 after a particle was released
                        // exactly at an output time -- or below the resolution of t, after AdvectionRK45's own clamp to
                        // RK45_min_dt): the reference's while-loop spins forever here (kernel.py:190).  A GPU must not:
                        // flag the particle (StatusCode.Error) instead.
                        c.state = PK_ERROR;
                        break;
                    }
                    p.x = padd(pf, p.x, p.dx);
                    p.y = padd(pf, p.y, p.dy);
                    p.z = padd(pf, p.z, p.dz);
                    p.t += p.dt;
                    p.dx = p.dy = p.dz = 0.0;
                    if (prm.rk45_mode) p.dt = p.next_dt;
                    steps++;
                }
                if (!prm.rk45_mode) p.dt = prm.dt0;                                 // :225-226
                if (c.state == PK_EVALUATE && p.t == endtime) c.state = PK_ENDOFLOOP;  // :229-230
            }
            O.t[i] = p.t;
            stp(O.z, i, p.z, pf);
            stp(O.y, i, p.y, pf);
            stp(O.x, i, p.x, pf);
            stp(O.dz, i, p.dz, pf);
            stp(O.dy, i, p.dy, pf);
            stp(O.dx, i, p.dx, pf);
            O.dt[i] = p.dt;
            if (P.next_dt) O.next_dt[i] = p.next_dt;
            O.state[i] = c.state;
            O.ei[i * ng] = c.ei0;
            if (ng > 1) O.ei[i * ng + 1] = c.ei1;
            if (ng > 2) O.ei[i * ng + 2] = c.ei2;
            if (ng > 3) O.ei[i * ng + 3] = c.ei3;
            O.iter[i] = body ? P.iter[i] : (int32_t)it;
            if (!body) note_error_iteration(a, c.state, it);
        }
    }
    steps = wave_sum(steps);
    attempts = wave_sum(attempts);
    paused = wave_sum(paused);
    if ((threadIdx.x & 63) == 0) {
        if (steps) atomicAdd(&a.counters->steps, steps);
        if (attempts) atomicAdd(&a.counters->attempts, attempts);
        if (paused) atomicAdd(&a.counters->paused, paused);
    }
}

#ifdef PK_USER_KERNELS
// A kernel of the list that rides along in a DEDICATED kernel (pk_fast_agrid.h / pk_fast_cgrid.h): a sampling-free recovery kernel or a
// user kernel that samples no field (the host checks what the module's kernels sample: pk_set_user_program), so its stage machine ends in
// stage 0.
PK_DEV void side_kernel(const KArgs& a, int kid, int kslot, int& state, bool pf, int64_t row, double& t, double& z, double& y, double& x,
                        double& dz, double& dy, double& dx, double& dt) {
    if (kid == PK_KERNEL_DELETE_ON_ERROR) {
        if (state >= PK_ERROR) state = PK_DELETE;
        return;
    }
    if (kid == PK_KERNEL_DELETE_OUT_OF_BOUNDS) {
        if (state == PK_ERROROUTOFBOUNDS || state == PK_ERRORTHROUGHSURFACE) state = PK_DELETE;
        return;
    }
    PState p;
    p.t = t; p.z = z; p.y = y; p.x = x; p.dz = dz; p.dy = dy; p.dx = dx; p.dt = dt;
    p.next_dt = 0.0;  // (kernels that touch next_dt do not ride here)
    p.id = a.p.particle_id[row];
    PCtx c;
    c.state = state;
    c.pf = pf;
    c.row = row;
    c.hz = c.hy = c.hx = c.ht = 0;
    c.hyx_valid = false;
    c.first_eval = 0u;
    c.u32 = c.v32 = c.oob = false;
    c.ei0 = c.ei1 = c.ei2 = c.ei3 = 0;
    c.it = 0u;
    c.klo = 0;
    c.zpos_f32 = false;
    KLocal L;
    Request rq;
    (void)user_prepare(a, kid - PK_KERNEL_USER0, 0, kslot, c, p, L, rq);
    t = p.t; z = p.z; y = p.y; x = p.x; dz = p.dz; dy = p.dy; dx = p.dx; dt = p.dt;
    state = c.state;
}
PK_DEV bool is_rk4_id(int kid) { return kid == PK_KERNEL_ADVECTION_RK4 || kid == PK_KERNEL_ADVECTION_RK4_3D; }

// The same inside the dedicated A-grid kernel, where a user kernel may also SAMPLE: the velocity field (eval_uvw_fast) or a scalar field of
// FastA::S (eval_scalar_fast; the host lists there what the module's kernels sample and checks that they share U's layout).  The stage
// machine of the kernel runs to its end here; every sample shares the search hints and the time / depth memo of the advection kernel.
template <class FT, bool PF, bool D3>
PK_DEV void side_kernel_fast(const KArgs& a, const FastTabs& ft, FCtx& fc, int kid, int kslot, int64_t row, unsigned it, double& t, double& z,
                             double& y, double& x, double& dz, double& dy, double& dx, double& dt) {
    if (kid == PK_KERNEL_DELETE_ON_ERROR) {
        if (fc.state >= PK_ERROR) fc.state = PK_DELETE;
        return;
    }
    if (kid == PK_KERNEL_DELETE_OUT_OF_BOUNDS) {
        if (fc.state == PK_ERROROUTOFBOUNDS || fc.state == PK_ERRORTHROUGHSURFACE) fc.state = PK_DELETE;
        return;
    }
    PState p;
    p.t = t; p.z = z; p.y = y; p.x = x; p.dz = dz; p.dy = dy; p.dx = dx; p.dt = dt;
    p.next_dt = 0.0;
    p.id = a.p.particle_id[row];
    PCtx c;
    c.state = fc.state;
    c.pf = PF;
    c.row = row;
    c.hz = c.hy = c.hx = c.ht = 0;
    c.hyx_valid = false;
    c.first_eval = 0u;
    c.u32 = c.v32 = c.oob = false;
    c.ei0 = c.ei1 = c.ei2 = c.ei3 = 0;
    c.it = it;
    c.klo = kslot * 1000;
    c.zpos_f32 = false;
    KLocal L;
    Request rq;
#pragma unroll 1
    for (int stage = 0; !user_prepare(a, kid - PK_KERNEL_USER0, stage, kslot, c, p, L, rq); stage++) {
        double u = 0.0, v = 0.0, w = 0.0;
        fc.state = c.state;
        if (rq.kind == RQ_SCALAR) {
            int slot = 0;
            for (int k = 1; k < a.fast.ns; k++) slot = a.fast.sfid[k] == rq.fidx ? k : slot;
            u = eval_scalar_fast<FT, PF>(a, ft, fc, slot, rq.t, rq.z, rq.y, rq.x, it, kslot * 1000 + stage);
        } else {
            eval_uvw_fast<FT, PF, D3>(a, ft, fc, rq.t, rq.z, rq.y, rq.x, PF && rq.f32, u, v, w, it, kslot * 1000 + stage);
        }
        c.state = fc.state;
        L.r[3] = u; L.r[4] = v; L.r[5] = w;
    }
    t = p.t; z = p.z; y = p.y; x = p.x; dz = p.dz; dy = p.dy; dx = p.dx; dt = p.dt;
    fc.state = c.state;
}
#endif

// ---- the headline kernel: AdvectionRK4 / AdvectionRK4_3D on a rectilinear A-grid with float64 coordinates -----------------
// Same step loop as advect_kernel (Kernel.execute, kernel.py:174-247) with the Runge-Kutta stages of _advection.py:42-75 written
// out around ONE evaluation site (pk_fast_agrid.h) instead of the prepare / consume stage machine: fewer loop-carried values,
// no kernel-local register array, particle dtype and dimensionality fixed at compile time.
#ifndef PK_MIN_WAVES_FAST
#define PK_MIN_WAVES_FAST 4
#endif
template <class FT, int PFM, bool D3>
__global__ void __launch_bounds__(256, PK_MIN_WAVES_FAST) advect_fast_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    FastTabs ft;
    {
        // {a[i], 1 / (a[i+1] - a[i])} tables of time | depth | lat | lon, staged once per workgroup (coalesced 16-byte copies)
        pk_tab2* s_tab = reinterpret_cast<pk_tab2*>(smem);
        const pk_tab2* g_tab = reinterpret_cast<const pk_tab2*>(a.fast.tab);
        for (int k = threadIdx.x; k < a.fast.lds_n; k += 256) s_tab[k] = g_tab[k];
        __syncthreads();
        ft.time = s_tab + a.fast.lds_time;
        ft.depth = s_tab + a.fast.lds_depth;
        ft.lat = s_tab + a.fast.lds_lat;
        ft.lon = s_tab + a.fast.lds_lon;
        ft.blk = s_tab + a.fast.lds_blk + threadIdx.x;  // (read only where lds_blk != 0)
        pin_scalars(ft, a.fast);
        ft.fl = fast_flags(a.fast) | ((a.win_lo > -INFINITY || a.win_hi < INFINITY) ? FA_WIN : 0u) | (a.prm.dt0 > 0 ? FA_FWD : 0u) |
                (a.prm.max_iters > 0 ? FA_MAXIT : 0u);
    }
    // the row index is re-derived where it is needed (entry and exit) instead of living in two registers across the step loop
    auto row = [&]() { return (int64_t)xcd_swizzle(blockIdx.x, gridDim.x) * 256 + threadIdx.x; };
    unsigned steps = 0, attempts = 0, paused = 0;  // per lane and launch: 32 bits are plenty
    if (row() < a.p.n) {
        int64_t i = row();
        const DParticles& P = a.p;
        const DPOut& O = a.po;
        const pk_exec_params& prm = a.prm;
        constexpr bool pf = PFM == 1;
        FCtx c;
        c.state = prm.reset_state ? PK_EVALUATE : P.state[i];  // kernel.py:188
        if (c.state == PK_EVALUATE) {
            unsigned it = prm.reset_state ? 0u : (unsigned)P.iter[i];
            fctx_init(c, PK_EVALUATE, P.ei[i * P.ngrids + a.fast.grid]);  // only the velocity grid's `ei` is touched
            double pt = P.t[i];
            double pz = ldp(P.z, i, pf), py = ldp(P.y, i, pf), px = ldp(P.x, i, pf);
            double pdz = ldp(P.dz, i, pf), pdy = ldp(P.dy, i, pf), pdx = ldp(P.dx, i, pf);
            double pdt = P.dt[i];
            const double endtime = prm.endtime;
            while (c.state == PK_EVALUATE) {  // :190 (no kernel of these programs sets Repeat)
                uint32_t fl = ft.fl;  // (FastTabs::fl: the step loop's yes / no questions as scalar bit tests, not as saved lane masks)
                asm volatile("" : "+s"(fl));
                const bool fwd = (fl & FA_FWD) != 0;                   // sign = 1 if dt0 > 0 else -1 (kernel.py:186)
                const double tte = fwd ? endtime - pt : -(endtime - pt);  // sign * (endtime - t), the sign of a zero included
                if (!(tte >= 0)) break;  // :193-197
                if ((fl & FA_MAXIT) && it >= (unsigned)prm.max_iters) break;  // pk_execute_rerun: stop where the reference raised
                double dtc;
                if (fwd) dtc = fmax(fmin(pdt, tte), 0.0);  // :200-203
                else dtc = fmin(fmax(pdt, -tte), 0.0);
                if (fl & FA_WIN) {  // field-slab streaming (some field streams through a ring of levels): step only inside the resident time window (advect_kernel)
                    const double t1 = pt + dtc;
                    const double lo = fmin(pt, t1), hi = fmax(pt, t1);
                    if (lo < a.win_lo || hi > a.win_hi) { paused = 1; break; }
                }
                it++;
                pdt = dtc;
                attempts++;
#ifdef PK_USER_KERNELS
                // a list with user kernels: [kernels before the advection kernel ..., AdvectionRK4(_3D), kernels after it ...]
                int adv = 0;
                for (; adv < prm.nk && !is_rk4_id(prm.kernels[adv]); adv++) {
                    attempts++;
                    side_kernel_fast<FT, pf, D3>(a, ft, c, prm.kernels[adv], adv, row(), it, pt, pz, py, px, pdz, pdy, pdx, pdt);
                }
#else
                constexpr int adv = 0;
#endif
                // AdvectionRK4(_3D), _advection.py:42-75: (u1 + 2*u2 + 2*u3 + u4) summed left to right
                double su = 0.0, sv = 0.0, sw = 0.0, lu = 0.0, lv = 0.0, lw = 0.0;
#pragma unroll 1
                for (int stage = 0; stage < 4; stage++) {
                    double st = pt, sz = pz, sy = py, sx = px;
                    if (stage > 0) {
                        const double cdt = stage == 3 ? 1.0 : 0.5;  // u*1.0 == u and 1.0*dt == dt exactly
                        sx = px + lu * cdt * pdt;
                        sy = py + lv * cdt * pdt;
                        if (D3) sz = pz + lw * cdt * pdt;
                        st = pt + cdt * pdt;
                    }
                    double u, v, w;
                    // stages 2 / 3 share t, stage 4 and stage 1 of the next step too: the odd evaluations keep their corner block, the even
                    // ones test it (pk_fast_agrid.h: FCtx::bei)
                    eval_uvw_fast<FT, pf, D3>(a, ft, c, st, sz, sy, sx, pf && stage == 0, u, v, w, it, adv * 1000 + stage, (stage & 1) ? 2 : 1);
                    if (stage == 0) { su = u; sv = v; sw = w; }
                    else if (stage == 3) { su = su + u; sv = sv + v; sw = sw + w; }
                    else { su = su + 2 * u; sv = sv + 2 * v; sw = sw + 2 * w; }
                    lu = u; lv = v; lw = w;
                }
                constexpr double sixth = 1.0 / 6.0;  // RN(1/6): su / 6.0 exactly (div_by_recip), or within an ulp (PK_FAST_LEAN: one multiplication)
                pdx = pstore(pf, pdx + (PK_FAST_LEAN ? su * sixth : div_by_recip(su, 6.0, sixth)) * pdt);
                pdy = pstore(pf, pdy + (PK_FAST_LEAN ? sv * sixth : div_by_recip(sv, 6.0, sixth)) * pdt);
                if (D3) pdz = pstore(pf, pdz + (PK_FAST_LEAN ? sw * sixth : div_by_recip(sw, 6.0, sixth)) * pdt);
                for (int k = adv + 1; k < prm.nk; k++) {  // the sampling-free recovery kernels that may follow (Delete*)
                    const int kid = prm.kernels[k];
                    attempts++;
#ifdef PK_USER_KERNELS
                    if (kid >= PK_KERNEL_USER0) {
                        side_kernel_fast<FT, pf, D3>(a, ft, c, kid, k, row(), it, pt, pz, py, px, pdz, pdy, pdx, pdt);
                        continue;
                    }
#endif
                    if (kid == PK_KERNEL_DELETE_ON_ERROR) {
                        if (c.state >= PK_ERROR) c.state = PK_DELETE;
                    } else if (c.state == PK_ERROROUTOFBOUNDS || c.state == PK_ERRORTHROUGHSURFACE) {
                        c.state = PK_DELETE;  // PK_KERNEL_DELETE_OUT_OF_BOUNDS
                    }
                }
                if (c.state == PK_EVALUATE || c.state == PK_SUCCESS) {  // :219-222 -> _position_update :108-120
                    if (tte > 0 && pt + pdt == pt) {  // dt == 0 before endtime: see advect_kernel
                        c.state = PK_ERROR;
                        break;
                    }
                    px = padd(pf, px, pdx);
                    py = padd(pf, py, pdy);
                    pz = padd(pf, pz, pdz);
                    pt += pdt;
                    pdx = pdy = pdz = 0.0;
                    steps++;
                }
                pdt = prm.dt0;                                                        // :225-226 (not RK45 mode)
                if (c.state == PK_EVALUATE && pt == endtime) c.state = PK_ENDOFLOOP;  // :229-230
            }
            i = row();
            asm volatile("" : "+v"(i));  // keep it re-derived: not hoisted above the loop
            O.t[i] = pt;
            stp(O.z, i, pz, pf);
            stp(O.y, i, py, pf);
            stp(O.x, i, px, pf);
            stp(O.dz, i, pdz, pf);
            stp(O.dy, i, pdy, pf);
            stp(O.dx, i, pdx, pf);
            O.dt[i] = pdt;
            if (P.next_dt && O.next_dt != P.next_dt) O.next_dt[i] = P.next_dt[i];
            O.state[i] = c.state;
            for (int g = 0; g < P.ngrids; g++) O.ei[i * P.ngrids + g] = g == a.fast.grid ? c.ei : P.ei[i * P.ngrids + g];
            O.iter[i] = (int32_t)it;
            note_error_iteration(a, c.state, it);
        }
    }
    const unsigned long long wsteps = wave_sum((unsigned long long)steps), wattempts = wave_sum((unsigned long long)attempts),
                             wpaused = wave_sum((unsigned long long)paused);
    if ((threadIdx.x & 63) == 0) {
        if (wsteps) atomicAdd(&a.counters->steps, wsteps);
        if (wattempts) atomicAdd(&a.counters->attempts, wattempts);
        if (wpaused) atomicAdd(&a.counters->paused, wpaused);
    }
}


// ---- AdvectionRK4 / AdvectionRK4_3D with CGrid_Velocity on a spherical curvilinear C-grid (BASELINE configs 3 / 4) -------------
// The step loop of advect_fast_kernel around the evaluation site of pk_fast_cgrid.h; one-wavefront workgroups, because the
// per-lane LDS slot (record of the particle's cell + its 12 staggered field values, ~15 KB per wavefront) is what bounds residency.
// Occupancy target and cell-cache placement (pk_fast_cgrid.h: CG_FV_REGS | CG_PXY_REGS) of the three kernels, chosen by measurement on
// config 3 / 5 (tools/ab_cgrid_occupancy.sh, profiles/r03_r_cgrid_occupancy.txt); overridable for A/B builds.
#ifndef PK_MIN_WAVES_CGRID
#define PK_MIN_WAVES_CGRID 3
#endif
#ifndef PK_MIN_WAVES_CGRID_RK45
#define PK_MIN_WAVES_CGRID_RK45 3
#endif
#ifndef PK_MIN_WAVES_CGRID_M1
#define PK_MIN_WAVES_CGRID_M1 3
#endif
#ifndef PK_CG_CACHE
#define PK_CG_CACHE 2
#endif
#ifndef PK_CG_CACHE_RK45
// round 6: CG_PXY_GLOBAL | CG_DMA | CG_PIN | CG_NO_TMEMO (pk_fast_cgrid.h).  Rounds 3-5 kept the corner coordinates in registers (2): 160 B
// / lane of scratch whose write-back was 5.4 of the 6.4 GB the kernel wrote per launch (state: 1.0 GB).  With the coordinates read from the
// table where they are used, the cell fetch by LDS-DMA, the hot scalars pinned and no time memo the allocator fits the 168 registers of 3
// waves per SIMD: 28 B of scratch, no spill store inside the step loop.  Kernel time is unchanged (13.8-13.9 ms either way, same box,
// alternating: profiles/r06c_rk45_variants.md) -- the kernel is bound by dependent latency at this residency, not by the spill traffic.
#define PK_CG_CACHE_RK45 60
#endif
#ifndef PK_CG_CACHE_M1
#define PK_CG_CACHE_M1 4
#endif
constexpr int CG_CACHE_RK4 = PK_CG_CACHE, CG_CACHE_RK45 = PK_CG_CACHE_RK45, CG_CACHE_M1 = PK_CG_CACHE_M1;
template <class FT, int PFM, bool D3, bool NE>
__global__ void __launch_bounds__(FC_LANES, PK_MIN_WAVES_CGRID) advect_cgrid_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const FastC& F = a.fastc;
    CgLds L;
    {
        pk_tab2* s_tab = reinterpret_cast<pk_tab2*>(smem);
        const pk_tab2* g_tab = reinterpret_cast<const pk_tab2*>(F.tab);
        for (int k = threadIdx.x; k < F.lds_n; k += FC_LANES) s_tab[k] = g_tab[k];
        __syncthreads();
        L.time = s_tab + F.lds_time;
        L.depth = s_tab + F.lds_depth;
        L.rec = smem + F.lds_rec + ((CG_CACHE_RK4 & CG_DMA) ? 2 * threadIdx.x : threadIdx.x);  // lane-private slots: no barrier needed
        L.fv = (void*)((FT*)(smem + F.lds_fv) + threadIdx.x);
        L.rec_w = reinterpret_cast<char*>(smem + F.lds_rec);
        L.fv_w = reinterpret_cast<char*>(smem + F.lds_fv);
        cg_pin<CG_CACHE_RK4>(L, F);
    }
    auto row = [&]() { return (int64_t)xcd_swizzle(blockIdx.x, gridDim.x) * FC_LANES + threadIdx.x; };
    unsigned steps = 0, attempts = 0, paused = 0;
    if (row() < a.p.n) {
        int64_t i = row();
        const DParticles& P = a.p;
        const DPOut& O = a.po;
        const pk_exec_params& prm = a.prm;
        constexpr bool pf = PFM == 1;
        CCtxT<FT, CG_CACHE_RK4> c;
        c.state = prm.reset_state ? PK_EVALUATE : P.state[i];  // kernel.py:188
        if (c.state == PK_EVALUATE) {
            unsigned it = prm.reset_state ? 0u : (unsigned)P.iter[i];
            {
                const int32_t ei0 = P.ei[i * P.ngrids + F.grid];
                int gy, gx;
                unravel_yx(kgrid(a, F.grid), (int64_t)ei0, gy, gx);  // the guess of the first search (index_search.py:269-274)
                cctx_init(c, PK_EVALUATE, ei0, gy, gx);
            }
            double pt = P.t[i];
            double pz = ldp(P.z, i, pf), py = ldp(P.y, i, pf), px = ldp(P.x, i, pf);
            double pdz = ldp(P.dz, i, pf), pdy = ldp(P.dy, i, pf), pdx = ldp(P.dx, i, pf);
            double pdt = P.dt[i];
            const double endtime = prm.endtime;
            const int sign = prm.dt0 > 0 ? 1 : -1;  // kernel.py:186
            const bool windowed = a.win_lo > -INFINITY || a.win_hi < INFINITY;
            while (c.state == PK_EVALUATE) {  // :190
                const double tte = sign * (endtime - pt);
                if (!(tte >= 0)) break;  // :193-197
                if (prm.max_iters > 0 && it >= (unsigned)prm.max_iters) break;  // pk_execute_rerun: stop where the reference raised
                double dtc;
                if (sign == 1) dtc = fmax(fmin(pdt, tte), 0.0);  // :200-203
                else dtc = fmin(fmax(pdt, -tte), 0.0);
                if (windowed) {  // field-slab streaming: step only inside the resident time window (advect_kernel)
                    const double t1 = pt + dtc;
                    const double lo = fmin(pt, t1), hi = fmax(pt, t1);
                    if (lo < a.win_lo || hi > a.win_hi) { paused = 1; break; }
                }
                it++;
                pdt = dtc;
                attempts++;
#ifdef PK_USER_KERNELS
                int adv = 0;  // a list with user kernels: [kernels before the advection kernel ..., AdvectionRK4(_3D), kernels after it ...]
                for (; adv < prm.nk && !is_rk4_id(prm.kernels[adv]); adv++) {
                    attempts++;
                    side_kernel(a, prm.kernels[adv], adv, c.state, pf, row(), pt, pz, py, px, pdz, pdy, pdx, pdt);
                }
#else
                constexpr int adv = 0;
#endif
                // AdvectionRK4(_3D), _advection.py:42-75: (u1 + 2*u2 + 2*u3 + u4) summed left to right
                double su = 0.0, sv = 0.0, sw = 0.0, lu = 0.0, lv = 0.0, lw = 0.0;
                if constexpr ((PK_CG_NEAR & 1) != 0) cg_home_sincos(c, py, px);  // the stage points lie near (py, px): sincos_near
#pragma unroll 1
                for (int stage = 0; stage < 4; stage++) {
                    double st = pt, sz = pz, sy = py, sx = px;
                    if (stage > 0) {
                        const double cdt = stage == 3 ? 1.0 : 0.5;  // u*1.0 == u and 1.0*dt == dt exactly
                        sx = px + lu * cdt * pdt;
                        sy = py + lv * cdt * pdt;
                        if (D3) sz = pz + lw * cdt * pdt;
                        st = pt + cdt * pdt;
                    }
                    double u, v, w;
                    eval_uvw_cgrid<FT, pf, D3, false, CG_CACHE_RK4, 1, (PK_CG_NEAR & 1) | (NE ? (PK_CG_NEAR & 2) : 0)>(a, L, c, st, sz, sy, sx, pf && stage == 0, u, v, w, it, adv * 1000 + stage, -1, py, px);
                    if (stage == 0) { su = u; sv = v; sw = w; }
                    else if (stage == 3) { su = su + u; sv = sv + v; sw = sw + w; }
                    else { su = su + 2 * u; sv = sv + 2 * v; sw = sw + 2 * w; }
                    lu = u; lv = v; lw = w;
                }
                constexpr double sixth = 1.0 / 6.0;  // RN(1/6): su / 6.0 exactly (div_by_recip), or within an ulp (PK_FAST_LEAN: one multiplication)
                pdx = pstore(pf, pdx + (PK_FAST_LEAN ? su * sixth : div_by_recip(su, 6.0, sixth)) * pdt);
                pdy = pstore(pf, pdy + (PK_FAST_LEAN ? sv * sixth : div_by_recip(sv, 6.0, sixth)) * pdt);
                if (D3) pdz = pstore(pf, pdz + (PK_FAST_LEAN ? sw * sixth : div_by_recip(sw, 6.0, sixth)) * pdt);
                for (int k = adv + 1; k < prm.nk; k++) {  // the sampling-free recovery kernels that may follow (Delete*)
                    const int kid = prm.kernels[k];
                    attempts++;
#ifdef PK_USER_KERNELS
                    if (kid >= PK_KERNEL_USER0) {
                        side_kernel(a, kid, k, c.state, pf, row(), pt, pz, py, px, pdz, pdy, pdx, pdt);
                        continue;
                    }
#endif
                    if (kid == PK_KERNEL_DELETE_ON_ERROR) {
                        if (c.state >= PK_ERROR) c.state = PK_DELETE;
                    } else if (c.state == PK_ERROROUTOFBOUNDS || c.state == PK_ERRORTHROUGHSURFACE) {
                        c.state = PK_DELETE;  // PK_KERNEL_DELETE_OUT_OF_BOUNDS
                    }
                }
                if (c.state == PK_EVALUATE || c.state == PK_SUCCESS) {  // :219-222 -> _position_update :108-120
                    if (tte > 0 && pt + pdt == pt) {  // dt == 0 before endtime: see advect_kernel
                        c.state = PK_ERROR;
                        break;
                    }
                    px = padd(pf, px, pdx);
                    py = padd(pf, py, pdy);
                    pz = padd(pf, pz, pdz);
                    pt += pdt;
                    pdx = pdy = pdz = 0.0;
                    steps++;
                }
                pdt = prm.dt0;                                                        // :225-226 (not RK45 mode)
                if (c.state == PK_EVALUATE && pt == endtime) c.state = PK_ENDOFLOOP;  // :229-230
            }
            i = row();
            asm volatile("" : "+v"(i));
            O.t[i] = pt;
            stp(O.z, i, pz, pf);
            stp(O.y, i, py, pf);
            stp(O.x, i, px, pf);
            stp(O.dz, i, pdz, pf);
            stp(O.dy, i, pdy, pf);
            stp(O.dx, i, pdx, pf);
            O.dt[i] = pdt;
            if (P.next_dt && O.next_dt != P.next_dt) O.next_dt[i] = P.next_dt[i];
            O.state[i] = c.state;
            for (int g = 0; g < P.ngrids; g++) O.ei[i * P.ngrids + g] = g == F.grid ? c.ei : P.ei[i * P.ngrids + g];
            O.iter[i] = (int32_t)it;
            note_error_iteration(a, c.state, it);
        }
    }
    const unsigned long long wsteps = wave_sum((unsigned long long)steps), wattempts = wave_sum((unsigned long long)attempts),
                             wpaused = wave_sum((unsigned long long)paused);
    if ((threadIdx.x & 63) == 0) {
        if (wsteps) atomicAdd(&a.counters->steps, wsteps);
        if (wattempts) atomicAdd(&a.counters->attempts, wattempts);
        if (wpaused) atomicAdd(&a.counters->paused, wpaused);
    }
}

// AdvectionRK45 (_advection.py:85-155) on the same evaluation site: the step loop of advect_kernel for the single-kernel RK45 program
// (Repeat loop, next_dt column, `dt = next_dt` of kernel.py:118-120), the Fehlberg stages written out like `prepare` does.  Every sample
// of these programs is guessed and float64 (the host checks it), so the float32-array products of the general program never arise.
template <class FT, int PFM, bool NE>
__global__ void __launch_bounds__(FC_LANES, PK_MIN_WAVES_CGRID_RK45) advect_cgrid_rk45_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const FastC& F = a.fastc;
    CgLds L;
    {
        pk_tab2* s_tab = reinterpret_cast<pk_tab2*>(smem);
        const pk_tab2* g_tab = reinterpret_cast<const pk_tab2*>(F.tab);
        for (int k = threadIdx.x; k < F.lds_n; k += FC_LANES) s_tab[k] = g_tab[k];
        __syncthreads();
        L.time = s_tab + F.lds_time;
        L.depth = s_tab + F.lds_depth;
        L.rec = smem + F.lds_rec + ((CG_CACHE_RK45 & CG_DMA) ? 2 * threadIdx.x : threadIdx.x);  // lane-private slots: no barrier needed
        L.fv = (void*)((FT*)(smem + F.lds_fv) + threadIdx.x);
        L.rec_w = reinterpret_cast<char*>(smem + F.lds_rec);
        L.fv_w = reinterpret_cast<char*>(smem + F.lds_fv);
        cg_pin<CG_CACHE_RK45>(L, F);
    }
    auto row = [&]() { return (int64_t)xcd_swizzle(blockIdx.x, gridDim.x) * FC_LANES + threadIdx.x; };
    unsigned steps = 0, attempts = 0, paused = 0;
    if (row() < a.p.n) {
        int64_t i = row();
        const DParticles& P = a.p;
        const DPOut& O = a.po;
        const pk_exec_params& prm = a.prm;
        constexpr bool pf = PFM == 1;
        CCtxT<FT, CG_CACHE_RK45> c;
        c.state = prm.reset_state ? PK_EVALUATE : P.state[i];  // kernel.py:188
        if (c.state == PK_EVALUATE) {
            unsigned it = prm.reset_state ? 0u : (unsigned)P.iter[i];
            {
                const int32_t ei0 = P.ei[i * P.ngrids + F.grid];
                int gy, gx;
                unravel_yx(kgrid(a, F.grid), (int64_t)ei0, gy, gx);
                cctx_init(c, PK_EVALUATE, ei0, gy, gx);
            }
            double pt = P.t[i];
            const double pz = ldp(P.z, i, pf);
            double py = ldp(P.y, i, pf), px = ldp(P.x, i, pf);
            double pdz = ldp(P.dz, i, pf), pdy = ldp(P.dy, i, pf), pdx = ldp(P.dx, i, pf);
            double pdt = P.dt[i], pnd = P.next_dt[i];
            double pzz = pz;
            const double endtime = prm.endtime;
            const int sign = prm.dt0 > 0 ? 1 : -1;
            const bool windowed = a.win_lo > -INFINITY || a.win_hi < INFINITY;
            while (c.state == PK_EVALUATE || c.state == PK_REPEAT) {  // :190
                const double tte = sign * (endtime - pt);
                if (!(tte >= 0)) break;  // :193-197
                if (prm.max_iters > 0 && it >= (unsigned)prm.max_iters) break;
                double dtc;
                if (sign == 1) dtc = fmax(fmin(pdt, tte), 0.0);  // :200-203
                else dtc = fmin(fmax(pdt, -tte), 0.0);
                if (windowed) {
                    const double t1 = pt + dtc;
                    const double lo = fmin(pt, t1), hi = fmax(pt, t1);
                    if (lo < a.win_lo || hi > a.win_hi) { paused = 1; break; }
                }
                it++;
                pdt = dtc;
                if constexpr ((PK_CG_NEAR_RK45 & 1) != 0) cg_home_sincos(c, py, px);  // once per iteration: rejected attempts start from the same point
                int sno = 0;  // samples taken in this iteration: the Repeat re-runs count on (pk_device.h: twe_listed)
                do {  // the Repeat loop of kernel.py:211-216
                    using namespace rk45c;
                    attempts++;
                    const double dt = pdt;
                    double u1 = 0, u2 = 0, u3 = 0, u4 = 0, u5 = 0, u6 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0;
#pragma unroll 1
                    for (int stage = 0; stage < 6; stage++) {
                        double sx = px, sy = py, st = pt;
                        switch (stage) {
                            case 1: sx = px + u1 * A00 * dt; sy = py + v1 * A00 * dt; st = pt + c0 * dt; break;
                            case 2: sx = px + (u1 * A10 + u2 * A11) * dt; sy = py + (v1 * A10 + v2 * A11) * dt; st = pt + c1 * dt; break;
                            case 3:
                                sx = px + (u1 * A20 + u2 * A21 + u3 * A22) * dt;
                                sy = py + (v1 * A20 + v2 * A21 + v3 * A22) * dt;
                                st = pt + c2 * dt;
                                break;
                            case 4:
                                sx = px + (u1 * A30 + u2 * A31 + u3 * A32 + u4 * A33) * dt;
                                sy = py + (v1 * A30 + v2 * A31 + v3 * A32 + v4 * A33) * dt;
                                st = pt + c3 * dt;
                                break;
                            case 5:
                                sx = px + (u1 * A40 + u2 * A41 + u3 * A42 + u4 * A43 + u5 * A44) * dt;
                                sy = py + (v1 * A40 + v2 * A41 + v3 * A42 + v4 * A43 + v5 * A44) * dt;
                                st = pt + c4 * dt;
                                break;
                            default: break;
                        }
                        double u, v, w;
                        eval_uvw_cgrid<FT, pf, false, false, CG_CACHE_RK45, PK_CG_HOPS, (PK_CG_NEAR_RK45 & 1) | (NE ? (PK_CG_NEAR_RK45 & 2) : 0)>(a, L, c, st, pzz, sy, sx, pf && stage == 0, u, v, w, it, sno + stage, -1, py, px);
                        switch (stage) {
                            case 0: u1 = u; v1 = v; break;
                            case 1: u2 = u; v2 = v; break;
                            case 2: u3 = u; v3 = v; break;
                            case 3: u4 = u; v4 = v; break;
                            case 4: u5 = u; v5 = v; break;
                            default: u6 = u; v6 = v; break;
                        }
                    }
                    const double sign_dt = (dt > 0) ? 1.0 : ((dt < 0) ? -1.0 : dt);  // np.sign
                    const double x_4th = (u1 * b40 + u2 * b41 + u3 * b42 + u4 * b43 + u5 * b44) * dt;
                    const double y_4th = (v1 * b40 + v2 * b41 + v3 * b42 + v4 * b43 + v5 * b44) * dt;
                    const double x_5th = (u1 * b50 + u2 * b51 + u3 * b52 + u4 * b53 + u5 * b54 + u6 * b55) * dt;
                    const double y_5th = (v1 * b50 + v2 * b51 + v3 * b52 + v4 * b53 + v5 * b54 + v6 * b55) * dt;
                    const double ex = x_5th - x_4th, ey = y_5th - y_4th;
                    const double kappa = sqrt(ex * ex + ey * ey);
                    const bool good = (kappa <= prm.rk45_tol) || (fabs(dt) <= fabs(prm.rk45_min_dt));
                    pdx = pstore(pf, pdx + (good ? x_5th : 0.0));
                    pdy = pstore(pf, pdy + (good ? y_5th : 0.0));
                    const bool increase = good && (kappa <= prm.rk45_tol / 10) && (fabs(dt * 2) <= fabs(prm.rk45_max_dt));
                    double next_dt = increase ? dt * 2 : dt;
                    if (fabs(next_dt) > fabs(prm.rk45_max_dt)) next_dt = prm.rk45_max_dt * sign_dt;
                    pnd = prm.next_dt_f32 ? (double)(float)next_dt : next_dt;
                    if (good) c.state = PK_EVALUATE;  // :146 overwrites any sampling error code
                    double ndt = good ? dt : dt / 2;
                    if (fabs(ndt) < fabs(prm.rk45_min_dt)) ndt = prm.rk45_min_dt * sign_dt;
                    pdt = ndt;
                    if (!good) c.state = PK_REPEAT;
                    sno += 6;
                } while (c.state == PK_REPEAT);
                for (int k = 1; k < prm.nk; k++) {  // the sampling-free recovery kernels that may follow (Delete*)
                    const int kid = prm.kernels[k];
                    attempts++;
                    if (kid == PK_KERNEL_DELETE_ON_ERROR) {
                        if (c.state >= PK_ERROR) c.state = PK_DELETE;
                    } else if (c.state == PK_ERROROUTOFBOUNDS || c.state == PK_ERRORTHROUGHSURFACE) {
                        c.state = PK_DELETE;
                    }
                }
                if (c.state == PK_EVALUATE || c.state == PK_SUCCESS) {  // :219-222 -> _position_update :108-120
                    if (tte > 0 && pt + pdt == pt) {
                        c.state = PK_ERROR;
                        break;
                    }
                    px = padd(pf, px, pdx);
                    py = padd(pf, py, pdy);
                    pzz = padd(pf, pzz, pdz);
                    pt += pdt;
                    pdx = pdy = pdz = 0.0;
                    pdt = pnd;  // kernel.py:118-120 (RK45 mode)
                    steps++;
                }
                if (c.state == PK_EVALUATE && pt == endtime) c.state = PK_ENDOFLOOP;  // :229-230
            }
            i = row();
            asm volatile("" : "+v"(i));
            O.t[i] = pt;
            stp(O.z, i, pzz, pf);
            stp(O.y, i, py, pf);
            stp(O.x, i, px, pf);
            stp(O.dz, i, pdz, pf);
            stp(O.dy, i, pdy, pf);
            stp(O.dx, i, pdx, pf);
            O.dt[i] = pdt;
            O.next_dt[i] = pnd;
            O.state[i] = c.state;
            for (int g = 0; g < P.ngrids; g++) O.ei[i * P.ngrids + g] = g == F.grid ? c.ei : P.ei[i * P.ngrids + g];
            O.iter[i] = (int32_t)it;
            note_error_iteration(a, c.state, it);
        }
    }
    const unsigned long long wsteps = wave_sum((unsigned long long)steps), wattempts = wave_sum((unsigned long long)attempts),
                             wpaused = wave_sum((unsigned long long)paused);
    if ((threadIdx.x & 63) == 0) {
        if (wsteps) atomicAdd(&a.counters->steps, wsteps);
        if (wattempts) atomicAdd(&a.counters->attempts, wattempts);
        if (wpaused) atomicAdd(&a.counters->paused, wpaused);
    }
}

// AdvectionDiffusionM1 (_advectiondiffusion.py:21-67) on the same evaluation site: Kh_zonal at x +- dres, UV, Kh_zonal at x, Kh_meridional
// at y +- dres and at y -- seven samples whose `ei` guesses chain through one another exactly like in the general program (a sample that
// starts in the cell of the point returns float64 (xsi, eta), one that has to move returns the float32-rounded ones of a hash hit:
// SearchMemo of pk_device.h is that same rule, spelled as a re-use).
template <class FT, int PFM, bool NE>
__global__ void __launch_bounds__(FC_LANES, PK_MIN_WAVES_CGRID_M1) advect_cgrid_m1_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const FastC& F = a.fastc;
    CgLds L;
    {
        pk_tab2* s_tab = reinterpret_cast<pk_tab2*>(smem);
        const pk_tab2* g_tab = reinterpret_cast<const pk_tab2*>(F.tab);
        for (int k = threadIdx.x; k < F.lds_n; k += FC_LANES) s_tab[k] = g_tab[k];
        __syncthreads();
        L.time = s_tab + F.lds_time;
        L.depth = s_tab + F.lds_depth;
        L.rec = smem + F.lds_rec + ((CG_CACHE_M1 & CG_DMA) ? 2 * threadIdx.x : threadIdx.x);  // lane-private slots: no barrier needed
        L.fv = (void*)((FT*)(smem + F.lds_fv) + threadIdx.x);
        L.rec_w = reinterpret_cast<char*>(smem + F.lds_rec);
        L.fv_w = reinterpret_cast<char*>(smem + F.lds_fv);
        cg_pin<CG_CACHE_M1>(L, F);
    }
    auto row = [&]() { return (int64_t)xcd_swizzle(blockIdx.x, gridDim.x) * FC_LANES + threadIdx.x; };
    unsigned steps = 0, attempts = 0, paused = 0;
    if (row() < a.p.n) {
        int64_t i = row();
        const DParticles& P = a.p;
        const DPOut& O = a.po;
        const pk_exec_params& prm = a.prm;
        constexpr bool pf = PFM == 1;
        CCtxT<FT, CG_CACHE_M1> c;
        c.state = prm.reset_state ? PK_EVALUATE : P.state[i];  // kernel.py:188
        if (c.state == PK_EVALUATE) {
            unsigned it = prm.reset_state ? 0u : (unsigned)P.iter[i];
            {
                const int32_t ei0 = P.ei[i * P.ngrids + F.grid];
                int gy, gx;
                unravel_yx(kgrid(a, F.grid), (int64_t)ei0, gy, gx);
                cctx_init(c, PK_EVALUATE, ei0, gy, gx);
            }
            double pt = P.t[i];
            double pz = ldp(P.z, i, pf), py = ldp(P.y, i, pf), px = ldp(P.x, i, pf);
            double pdz = ldp(P.dz, i, pf), pdy = ldp(P.dy, i, pf), pdx = ldp(P.dx, i, pf);
            double pdt = P.dt[i];
            const int64_t pid = P.particle_id[i];
            const double endtime = prm.endtime;
            const int sign = prm.dt0 > 0 ? 1 : -1;
            const bool windowed = a.win_lo > -INFINITY || a.win_hi < INFINITY;
            const double dres = prm.dres;
            while (c.state == PK_EVALUATE) {  // :190
                const double tte = sign * (endtime - pt);
                if (!(tte >= 0)) break;  // :193-197
                if (prm.max_iters > 0 && it >= (unsigned)prm.max_iters) break;
                double dtc;
                if (sign == 1) dtc = fmax(fmin(pdt, tte), 0.0);  // :200-203
                else dtc = fmin(fmax(pdt, -tte), 0.0);
                if (windowed) {
                    const double t1 = pt + dtc;
                    const double lo = fmin(pt, t1), hi = fmax(pt, t1);
                    if (lo < a.win_lo || hi > a.win_hi) { paused = 1; break; }
                }
                it++;
                pdt = dtc;
                attempts++;
                // seven samples at one call site (_advectiondiffusion.py:44-57): Kxp1, Kxm1, UV, khz, Kyp1, Kym1, khm
                double Kxp1 = 0, Kxm1 = 0, khz = 0, Kyp1 = 0, Kym1 = 0, khm = 0, u = 0, v = 0;
                cg_home_sincos(c, py, px);
#pragma unroll 1
                for (int stage = 0; stage < 7; stage++) {
                    double sx = px, sy = py;
                    int sk = stage < 4 ? 0 : 1;
                    switch (stage) {
                        case 0: sx = padd(pf, px, dres); break;
                        case 1: sx = psub(pf, px, dres); break;
                        case 2: sk = -1; break;
                        case 4: sy = padd(pf, py, dres); break;
                        case 5: sy = psub(pf, py, dres); break;
                        default: break;
                    }
                    double r0, r1, r2;
                    eval_uvw_cgrid<FT, pf, false, true, CG_CACHE_M1, 1, (PK_CG_NEAR_M1 & 1) | (NE ? (PK_CG_NEAR_M1 & 2) : 0)>(a, L, c, pt, pz, sy, sx, pf, r0, r1, r2, it, stage, sk, py, px);
                    switch (stage) {
                        case 0: Kxp1 = r0; break;
                        case 1: Kxm1 = r0; break;
                        case 2: u = r0; v = r1; break;
                        case 3: khz = r0; break;
                        case 4: Kyp1 = r0; break;
                        case 5: Kym1 = r0; break;
                        default: khm = r0; break;
                    }
                }
                // prepare(PK_KERNEL_ADVECTIONDIFFUSION_M1) of the general program, spherical mesh
                Kxp1 = m2_to_deg2_zonal(pf, Kxp1, py, F.deg2m);
                Kxm1 = m2_to_deg2_zonal(pf, Kxm1, py, F.deg2m);
                khz = m2_to_deg2_zonal(pf, khz, py, F.deg2m);
                Kyp1 = m2_to_deg2_merid(Kyp1, F.deg2m);
                Kym1 = m2_to_deg2_merid(Kym1, F.deg2m);
                khm = m2_to_deg2_merid(khm, F.deg2m);
                const double dKdx = (Kxp1 - Kxm1) / (2 * dres), dKdy = (Kyp1 - Kym1) / (2 * dres);
                const double bx = sqrt(2 * khz), by = sqrt(2 * khm);
                double z0, z1;
                normal_pair(prm.seed, 0, pid, pt, z0, z1);
                const double sq = sqrt(fabs(pdt));
                const double dWx = sq * z0, dWy = sq * z1;
                pdx = pstore(pf, pdx + (u * pdt + 0.5 * dKdx * (dWx * dWx + pdt) + bx * dWx));  // :66-67
                pdy = pstore(pf, pdy + (v * pdt + 0.5 * dKdy * (dWy * dWy + pdt) + by * dWy));
                for (int k = 1; k < prm.nk; k++) {  // the sampling-free recovery kernels that may follow (Delete*)
                    const int kid = prm.kernels[k];
                    attempts++;
                    if (kid == PK_KERNEL_DELETE_ON_ERROR) {
                        if (c.state >= PK_ERROR) c.state = PK_DELETE;
                    } else if (c.state == PK_ERROROUTOFBOUNDS || c.state == PK_ERRORTHROUGHSURFACE) {
                        c.state = PK_DELETE;
                    }
                }
                if (c.state == PK_EVALUATE || c.state == PK_SUCCESS) {  // :219-222 -> _position_update :108-120
                    if (tte > 0 && pt + pdt == pt) {
                        c.state = PK_ERROR;
                        break;
                    }
                    px = padd(pf, px, pdx);
                    py = padd(pf, py, pdy);
                    pz = padd(pf, pz, pdz);
                    pt += pdt;
                    pdx = pdy = pdz = 0.0;
                    steps++;
                }
                pdt = prm.dt0;                                                        // :225-226 (not RK45 mode)
                if (c.state == PK_EVALUATE && pt == endtime) c.state = PK_ENDOFLOOP;  // :229-230
            }
            i = row();
            asm volatile("" : "+v"(i));
            O.t[i] = pt;
            stp(O.z, i, pz, pf);
            stp(O.y, i, py, pf);
            stp(O.x, i, px, pf);
            stp(O.dz, i, pdz, pf);
            stp(O.dy, i, pdy, pf);
            stp(O.dx, i, pdx, pf);
            O.dt[i] = pdt;
            if (P.next_dt && O.next_dt != P.next_dt) O.next_dt[i] = P.next_dt[i];
            O.state[i] = c.state;
            for (int g = 0; g < P.ngrids; g++) O.ei[i * P.ngrids + g] = g == F.grid ? c.ei : P.ei[i * P.ngrids + g];
            O.iter[i] = (int32_t)it;
            note_error_iteration(a, c.state, it);
        }
    }
    const unsigned long long wsteps = wave_sum((unsigned long long)steps), wattempts = wave_sum((unsigned long long)attempts),
                             wpaused = wave_sum((unsigned long long)paused);
    if ((threadIdx.x & 63) == 0) {
        if (wsteps) atomicAdd(&a.counters->steps, wsteps);
        if (wattempts) atomicAdd(&a.counters->attempts, wattempts);
        if (wpaused) atomicAdd(&a.counters->paused, wpaused);
    }
}

// fast C-grid programs: one TU defines launch_cgrid (field dtype x particle dtype x 2-D / 3-D)
void launch_cgrid(int field_f32, int particles_f32, int d3, const KArgs& a, int64_t n, size_t lds_bytes, hipStream_t stream);
void launch_cgrid_rk45(int field_f32, int particles_f32, const KArgs& a, int64_t n, size_t lds_bytes, hipStream_t stream);
void launch_cgrid_m1(int field_f32, int particles_f32, const KArgs& a, int64_t n, size_t lds_bytes, hipStream_t stream);

// One translation unit per program (compiled in parallel) defines launch_program<PROG>.
// key bits: field f32 | curvilinear | C-grid ; lds: coordinate vectors staged in LDS
template <int PROG>
void launch_program(int field_f32, int curvilinear, int interp, int lds, const KArgs& a, dim3 grid, size_t lds_bytes,
                    hipStream_t stream);

// fast single-kernel programs (pk_fast_agrid.h): one TU per program defines launch_fast<PROG>
template <int PROG>
void launch_fast(int field_f32, int particles_f32, const KArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream);

#define PK_LAUNCH_FAST_CASE(FT, PF) \
    hipLaunchKernelGGL((advect_fast_kernel<FT, PF, KIDV == PK_KERNEL_ADVECTION_RK4_3D>), grid, dim3(256), lds_bytes, stream, a)
#define PK_DEFINE_LAUNCH_FAST(PROGV, KID_)                                                                              \
    template <>                                                                                                         \
    void launch_fast<PROGV>(int field_f32, int particles_f32, const KArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream) { \
        constexpr int KIDV = KID_;                                                                                      \
        if (field_f32) {                                                                                                \
            if (particles_f32) PK_LAUNCH_FAST_CASE(float, 1); else PK_LAUNCH_FAST_CASE(float, 0);                       \
        } else {                                                                                                        \
            if (particles_f32) PK_LAUNCH_FAST_CASE(double, 1); else PK_LAUNCH_FAST_CASE(double, 0);                     \
        }                                                                                                               \
    }

// PK_PRINT_OCCUPANCY=1: what the runtime will co-schedule of the chosen instantiation (workgroups per CU at this LDS size)
inline bool print_occupancy() {
    static const bool v = getenv("PK_PRINT_OCCUPANCY") != nullptr;
    return v;
}
#define PK_LAUNCH_CASE(FT, KD, IN, LD)                                                                                          \
    do {                                                                                                                        \
        if (print_occupancy()) {                                                                                                \
            int nb = 0;                                                                                                         \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, advect_kernel<FT, KD, IN, KIDV, LD, TYPEDV>, wg_size(KD, LD), lds_bytes); \
            fprintf(stderr, "[pk] advect_kernel<%s,kind %d,interp %d,kid %d> wg %d lds %zu B: %d workgroups / CU\n", sizeof(FT) == 4 ? "f32" : "f64", KD, IN, KIDV, wg_size(KD, LD), (size_t)lds_bytes, nb); \
        }                                                                                                                       \
        hipLaunchKernelGGL((advect_kernel<FT, KD, IN, KIDV, LD, TYPEDV>), dim3((unsigned)((a.p.n + wg_size(KD, LD) - 1) / wg_size(KD, LD))), \
                           dim3(wg_size(KD, LD)), lds_bytes, stream, a);                                                        \
    } while (0)

// single-kernel programs require LDS staging (the host falls back to the generic program otherwise)
// interp: 0 XLinear_Velocity, 1 CGrid_Velocity, 2 slip (XFreeslip / XPartialslip, told apart by prm.interp_uv)
#define PK_LAUNCH_KEYS(LD)                                     \
    switch (key) {                                             \
        case 0: PK_LAUNCH_CASE(double, 0, 0, LD); break;       \
        case 1: PK_LAUNCH_CASE(double, 0, 1, LD); break;       \
        case 2: PK_LAUNCH_CASE(double, 0, 2, LD); break;       \
        case 3: PK_LAUNCH_CASE(double, 1, 0, LD); break;       \
        case 4: PK_LAUNCH_CASE(double, 1, 1, LD); break;       \
        case 5: PK_LAUNCH_CASE(double, 1, 2, LD); break;       \
        case 6: PK_LAUNCH_CASE(float, 0, 0, LD); break;        \
        case 7: PK_LAUNCH_CASE(float, 0, 1, LD); break;        \
        case 8: PK_LAUNCH_CASE(float, 0, 2, LD); break;        \
        case 9: PK_LAUNCH_CASE(float, 1, 0, LD); break;        \
        case 10: PK_LAUNCH_CASE(float, 1, 1, LD); break;       \
        case 11: PK_LAUNCH_CASE(float, 1, 2, LD); break;       \
    }

#define PK_DEFINE_LAUNCH_PROGRAM(PROGV, KID_, WITH_NOLDS, TYPED_)                                                    \
    template <>                                                                                                      \
    void launch_program<PROGV>(int field_f32, int curvilinear, int interp, int lds, const KArgs& a, dim3 grid,       \
                               size_t lds_bytes, hipStream_t stream) {                                               \
        constexpr int KIDV = KID_;                                                                                   \
        constexpr bool TYPEDV = TYPED_;                                                                              \
        const int ik = interp >= 2 ? 2 : interp;                                                                     \
        const int key = (field_f32 ? 6 : 0) + (curvilinear ? 3 : 0) + ik;                                            \
        if (lds || !(WITH_NOLDS)) {                                                                                  \
            PK_LAUNCH_KEYS(true)                                                                                     \
        } else if (WITH_NOLDS) {                                                                                     \
            PK_LAUNCH_KEYS(!(WITH_NOLDS))                                                                            \
        }                                                                                                            \
    }

}  // namespace pk
