/* fast_agrid_cpu.c -- TEST / MEASUREMENT INFRASTRUCTURE: the "fair native CPU" leg of bench.py's cpu_baseline (kind "port").
 *
 * parcels_oracle.c is a checker: it carries NumPy's dtype emulation, searches every coordinate by bisection from scratch and keeps
 * its context in structs -- 9e4 particle-steps/s per thread, slower than the NumPy reference's own batch loop.  That is no yardstick
 * for a GPU kernel.  This file restates ONLY the headline workload -- AdvectionRK4 (_advection.py:42-55) on a rectilinear A-grid with
 * float64 coordinates, fields and particles (BASELINE config 2): Kernel.execute's loop (kernel.py:190-230), _search_1d_array
 * (index_search.py:20-62), _search_time_index (:65-91), XLinear (_xinterpolators.py:112-153), XLinear_Velocity (:169-190) -- the way
 * one would write it for a CPU: the cell of the previous evaluation as search hint, no dtype tags, the two fields gathered with one
 * set of offsets, OpenMP over (cell-sorted) particles.  Same operations in the same order as parcels_oracle.c for this case:
 * tests/test_oracle_fast_cpu.py holds it to the oracle BIT FOR BIT, so it inherits the oracle's pin to the reference.
 *
 * Particles that would leave the domain or the time interval make the call return 1 (the caller then uses the general oracle):
 * the bench workload has none.  Not part of the product: only bench.py's cpu_baseline leg and tests/ may call it.
 */
#include <math.h>
#include <stdint.h>

static const double DEG2RAD = 3.14159265358979323846 / 180.0; /* npy_deg2rad */

/* clip(searchsorted(arr, x, "left") - 1, 0, n - 2) starting from the hint (particles move less than a cell per evaluation) */
static inline int cell_of(const double* arr, int n, double x, int i) {
    while (i < n - 2 && arr[i + 1] < x) i++;
    while (i > 0 && !(arr[i] < x)) i--;
    return i;
}

typedef struct {
    const double *lon, *lat, *depth, *time, *U, *V;
    int nx, ny, nz, nt, spherical;
    double deg2m, tlen;
    int64_t sy, sz, st;
} grid_t;

/* VectorField.eval for UV.  Returns 0, or 1 when the point is outside the grid / the time interval (NaN included). */
static inline int eval_uv(const grid_t* g, double t, double z, double y, double x, int* ht, int* hz, int* hy, int* hx, double* u, double* v) {
    int ti = 0, zi = 0, yi = 0, xi = 0;
    double tau = 0.0, zeta = 0.0, eta = 0.0, xsi = 0.0;
    if (g->nt > 1) {
        if (!(0 <= t) || !(t <= g->tlen)) return 1;
        ti = *ht = cell_of(g->time, g->nt, t, *ht);
        tau = (t - g->time[ti]) / (g->time[ti + 1] - g->time[ti]);
    }
    if (g->nz > 1) {
        if (!(z >= g->depth[0]) || !(z <= g->depth[g->nz - 1])) return 1;
        zi = *hz = cell_of(g->depth, g->nz, z, *hz);
        zeta = (z - g->depth[zi]) / (g->depth[zi + 1] - g->depth[zi]);
    }
    if (g->ny > 1) {
        if (!(y >= g->lat[0]) || !(y <= g->lat[g->ny - 1])) return 1;
        yi = *hy = cell_of(g->lat, g->ny, y, *hy);
        eta = (y - g->lat[yi]) / (g->lat[yi + 1] - g->lat[yi]);
    }
    if (g->nx > 1) {
        if (!(x >= g->lon[0]) || !(x <= g->lon[g->nx - 1])) return 1;
        xi = *hx = cell_of(g->lon, g->nx, x, *hx);
        xsi = (x - g->lon[xi]) / (g->lon[xi + 1] - g->lon[xi]);
    }
    const int lenT = tau > 0, lenZ = !(zeta <= 0);
    const int64_t t0 = ti * g->st, t1 = (ti + 1 < g->nt ? ti + 1 : g->nt - 1) * g->st;
    const int64_t z0 = zi * g->sz, z1 = (zi + 1 < g->nz ? zi + 1 : g->nz - 1) * g->sz;
    const int64_t y0 = yi * g->sy, y1 = (yi + 1 < g->ny ? yi + 1 : g->ny - 1) * g->sy;
    const int64_t x0 = xi, x1 = xi + 1 < g->nx ? xi + 1 : g->nx - 1;
    const double omt = 1 - tau, omz = 1 - zeta, omx = 1 - xsi, ome = 1 - eta;
    double r[2];
    const double* F[2] = {g->U, g->V};
    for (int f = 0; f < 2; f++) {
        const double* d = F[f];
        double c[2][2];
        for (int iy = 0; iy < 2; iy++)
            for (int ix = 0; ix < 2; ix++) {
                const int64_t o = (iy ? y1 : y0) + (ix ? x1 : x0);
                double a = d[t0 + z0 + o];
                if (lenT) a = a * omt + d[t1 + z0 + o] * tau;
                if (lenZ) {
                    double b = d[t0 + z1 + o];
                    if (lenT) b = b * omt + d[t1 + z1 + o] * tau;
                    a = a * omz + b * zeta;
                }
                c[iy][ix] = a;
            }
        r[f] = (omx * ome) * c[0][0] + (xsi * ome) * c[0][1] + (omx * eta) * c[1][0] + (xsi * eta) * c[1][1];
    }
    if (g->spherical) { /* _xinterpolators.py:183-187 */
        r[0] /= g->deg2m * cos(y * DEG2RAD);
        r[1] /= g->deg2m;
    }
    *u = r[0];
    *v = r[1];
    return (r[0] != r[0]) || (r[1] != r[1]);
}

/* Kernel.execute(pset, endtime, dt) with kernels = [AdvectionRK4]: every particle from its own t to endtime.
 * state_out: 1 = EndofLoop (kernel.py:229-230), 10 = still Evaluate (t past endtime on entry), -1 = left the domain (unsupported here).
 * Returns the number of particles with state -1. */
int64_t pf_rk4_agrid(const double* lon, int nx, const double* lat, int ny, const double* depth, int nz, const double* time, int nt,
                     const double* U, const double* V, int spherical, double deg2m, int64_t n, double* pt, const double* pz, double* py,
                     double* px, int32_t* state_out, double dt0, double endtime, int nthreads, int64_t* steps_out) {
    grid_t g = {lon, lat, depth, time, U, V, nx, ny, nz, nt, spherical, deg2m, nt > 1 ? time[nt - 1] - time[0] : 0.0,
                (int64_t)nx, (int64_t)ny * nx, (int64_t)nz * ny * nx};
    const int sign = dt0 > 0 ? 1 : -1;
    int64_t steps = 0, bad = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : steps, bad) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; i++) {
        double t = pt[i], y = py[i], x = px[i], dt = dt0;
        const double z = pz[i];
        int ht = 0, hz = 0, hy = 0, hx = 0, st = 10, err = 0;
        while (st == 10) {
            const double tte = sign * (endtime - t);
            if (!(tte >= 0)) break;
            dt = sign == 1 ? fmax(fmin(dt, tte), 0) : fmin(fmax(dt, -tte), 0); /* kernel.py:199-203 */
            double u1, v1, u2, v2, u3, v3, u4, v4;
            err = eval_uv(&g, t, z, y, x, &ht, &hz, &hy, &hx, &u1, &v1);
            const double x1 = x + u1 * 0.5 * dt, y1 = y + v1 * 0.5 * dt;
            err = err || eval_uv(&g, t + 0.5 * dt, z, y1, x1, &ht, &hz, &hy, &hx, &u2, &v2);
            const double x2 = x + u2 * 0.5 * dt, y2 = y + v2 * 0.5 * dt;
            err = err || eval_uv(&g, t + 0.5 * dt, z, y2, x2, &ht, &hz, &hy, &hx, &u3, &v3);
            const double x3 = x + u3 * dt, y3 = y + v3 * dt;
            err = err || eval_uv(&g, t + dt, z, y3, x3, &ht, &hz, &hy, &hx, &u4, &v4);
            if (err || (tte > 0 && t + dt == t)) { st = -1; break; }
            const double dx = 0.0 + (u1 + 2 * u2 + 2 * u3 + u4) / 6.0 * dt, dy = 0.0 + (v1 + 2 * v2 + 2 * v3 + v4) / 6.0 * dt;
            x += dx; /* _position_update (kernel.py:108-120) */
            y += dy;
            t += dt;
            steps++;
            dt = dt0;
            if (t == endtime) st = 1;
        }
        pt[i] = t; py[i] = y; px[i] = x;
        state_out[i] = st;
        bad += st == -1;
    }
    if (steps_out) *steps_out = steps;
    return bad;
}
