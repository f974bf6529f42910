"""CPU: the Python -> device-kernel translator of parcels_amd/jit.py, checked WITHOUT a GPU.  The C++ the translator emits for a kernel
is plain scalar code over `p.*`, `c.state`, the user-Variable columns and its locals, so it also compiles for the host: a 40-line shim
of the device structs (PState, PCtx, KLocal, Request), g++ -O2 -ffp-contract=off, a loop over the particles that plays the stage machine
-- a field sample is answered from arrays the test supplies -- and the result is compared BIT FOR BIT with NumPy running the same Python
function on the same columns through HostParticles (the object a Python kernel receives on the host path).  This pins the translator's
static NumPy semantics (NEP 50 promotion, in-place casts, true division, truncating stores, %, fmod, minimum / maximum with NaN, masked
stores ...) in the suite that needs no GPU; tests/test_gpu_jit_kernels.py repeats it end to end on the device."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import parcels_amd as pa
from parcels_amd import StatusCode, jit
from parcels_amd.hostkernels import HostParticles

SHIM = r"""
#include <stdint.h>
#include <math.h>
#include <string.h>
#define PK_DEV static inline
enum { RQ_UV = 0, RQ_UVW = 1, RQ_SCALAR = 2 };
enum { PK_ERROR = 50 };
struct PState { double t, z, y, x, dz, dy, dx, dt, next_dt; int64_t id; };
struct PCtx { int state; bool pf; int64_t row; int32_t ei0, ei1, ei2, ei3; };
struct Request { int kind, fidx; double t, z, y, x; bool f32; };
struct DParticles { void* extra[4]; };
struct KArgs { DParticles p; };
struct PkUserLocals {
%(decl)s
};
struct KLocal { double r[14]; PkUserLocals ul; };
PK_DEV bool user_prepare(const KArgs& a, int uk, int stage, int kslot, PCtx& c, PState& p, KLocal& L, Request& rq) {
    switch (uk) {
        case 0: {
%(body)s
        }
        default: return true;
    }
}
// cols: t z y x dz dy dx dt (double each, spatial ones hold float values when pf), state (int32), id (int64), 4 extra columns;
// samples[(fk * 3 + j) * n + i]: component j of field fk = kind * 16 + fidx for particle i; a sample also does what the library's does to the
// particle: state = max(state, sstate[fk * n + i]) and ei0 = sei[fk * n + i]; req[(k * 6 + j) * n + i]: t, z, y, x, f32 flag, kind * 100 + fidx
// of the k-th request of the kernel
extern "C" void run(int64_t n, int pf, double* t, double* z, double* y, double* x, double* dz, double* dy, double* dx, double* dt,
                    int32_t* state, int64_t* id, void* e0, void* e1, void* e2, void* e3, const double* samples, int32_t* nsamples,
                    const int32_t* sstate, const int32_t* sei, int32_t* ei0, double* req, uint8_t* asked, const int32_t* positional) {
    KArgs a;
    *nsamples = 0;
    a.p.extra[0] = e0; a.p.extra[1] = e1; a.p.extra[2] = e2; a.p.extra[3] = e3;
    for (int64_t i = 0; i < n; i++) {
        PState p = {t[i], z[i], y[i], x[i], dz[i], dy[i], dx[i], dt[i], 0.0, id[i]};
        PCtx c = {state[i], pf != 0, i, ei0[i], 0, 0, 0};
        KLocal L;
        memset(&L, 0, sizeof(L));
        Request rq;
        int k = 0;
        for (int stage = 0; !user_prepare(a, 0, stage, 0, c, p, L, rq); stage++) {
            // which sample of the kernel this is: the stage, or -- in a kernel whose samples are conditional -- its own counter
            k = %(ordinal)s;
            const int fk = rq.kind * 16 + rq.fidx;  // the field asked for: values and side effects are the FIELD's, whichever sample of the kernel this is
            for (int j = 0; j < 3; j++)
                L.r[3 + j] = positional[fk] ? (((rq.x * 1.25 + rq.y * 0.5) - rq.z * 0.25) + rq.t * 0.001) * (j + 1) : samples[((int64_t)fk * 3 + j) * n + i];
            const double r6[6] = {rq.t, rq.z, rq.y, rq.x, rq.f32 ? 1.0 : 0.0, (double)(rq.kind * 100 + rq.fidx)};
            for (int j = 0; j < 6; j++) req[((int64_t)k * 6 + j) * n + i] = r6[j];
            asked[(int64_t)k * n + i] = 1;
            if (sstate[(int64_t)fk * n + i] > c.state) c.state = sstate[(int64_t)fk * n + i];
            c.ei0 = sei[(int64_t)fk * n + i];
            if (k + 1 > *nsamples) *nsamples = k + 1;
        }
        ei0[i] = c.ei0;
        t[i] = p.t; z[i] = p.z; y[i] = p.y; x[i] = p.x; dz[i] = p.dz; dy[i] = p.dy; dx[i] = p.dx; dt[i] = p.dt; state[i] = c.state;
    }
}
"""


class _FakeField:
    """Answers a sample from arrays, logs the sample point, and does to the particles what the library's sampling does (when it is handed
    them): raises the state to `sstate` and writes `ei`."""

    def __init__(self, values, log=None, effects=None, positional=False, key=0):
        self.key = key  # kind * 16 + field id, as the shim computes it from a request
        self.values = values  # list of component arrays (1 for a scalar field, 2 / 3 for UV / UVW)
        self.positional = positional  # the value is a function of the sample point instead (samples without particles on a sub-selection)
        self.log = log if log is not None else []
        self.effects = effects  # (sstate[k], sei[k]) per sample of the run, or None

    def __getitem__(self, key):
        particles = key[4] if isinstance(key, tuple) and len(key) == 5 else (None if isinstance(key, tuple) else key)
        if particles is not None:
            rows = vars(particles).get("_rows")
            if rows is None:  # the reference's ParticleSetView: a boolean mask over the particle set
                rows = np.flatnonzero(vars(particles)["_index"])
        else:
            rows = np.arange(len(self.values[0])) if not self.positional else None  # (positional: whichever particles the points belong to)
        t, z, y, x = key[:4] if isinstance(key, tuple) else (particles.t, particles.z, particles.y, particles.x)
        k = len(self.log)
        self.log.append({"field": getattr(self, "name", None), "rows": np.asarray(rows),"t": np.asarray(t, dtype=np.float64), "z": np.asarray(z, dtype=np.float64), "y": np.asarray(y, dtype=np.float64),
                         "x": np.asarray(x, dtype=np.float64), "f32": np.asarray(y).dtype == np.float32, "attached": particles is not None, "implicit": not isinstance(key, tuple)})
        if particles is not None and self.effects is not None:
            sstate, sei = self.effects
            particles.state = np.maximum(np.asarray(particles.state), sstate[self.key][rows])
            vars(particles)["_data"]["ei"][rows, 0] = sei[self.key][rows]
        if self.positional:
            e = self.log[-1]
            base = ((e["x"] * 1.25 + e["y"] * 0.5) - e["z"] * 0.25) + e["t"] * 0.001
            out = tuple(base * (j + 1) for j in range(len(self.values)))
        else:
            out = tuple(v[rows] for v in self.values)
        return out[0] if len(out) == 1 else out


class _FakeFieldSet:
    def __init__(self, context, fields):
        self.context = dict(context)
        self.fields = fields

    def __getattr__(self, name):
        d = self.__dict__
        if name in d.get("fields", {}):
            return d["fields"][name]
        if name in d.get("context", {}):
            return d["context"][name]
        raise AttributeError(name)


def _columns(pclass, n, seed, finite=False):
    rng = np.random.default_rng(seed)
    data = {}
    for v in pclass.variables:
        dt = np.dtype(v.dtype)
        if v.name == "ei":
            data["ei"] = np.zeros((n, 1), np.int32)
        elif v.name == "state":
            data["state"] = rng.choice([int(StatusCode.Evaluate), int(StatusCode.Success), 60, 61, 70], size=n).astype(np.int32)
        elif v.name == "particle_id":
            data["particle_id"] = np.arange(n, dtype=np.int64) * 7 + 3
        elif dt.kind == "f":
            a = rng.normal(scale=3.0, size=n)
            a[rng.random(n) < 0.05] = 0.0
            if v.name not in ("t", "dt", "x", "y", "z") and not finite:
                a[rng.random(n) < 0.03] = np.nan
            data[v.name] = a.astype(dt)
        else:
            data[v.name] = rng.integers(-5, 9, size=n).astype(dt)
    data["dt"] = np.full(n, 600.0)
    data.setdefault("ei", np.zeros((n, 1), np.int32))
    return data


NKEY = 48  # kind (UV 0, UVW 1, scalar 2) * 16 + field id


def _run_translated(func, pclass, fieldset, data, var_slot, field_ids, samples, tmp_path, effects=None, positional=None, nord=1):
    """samples: [NKEY, 3, n] values per field key, effects: ([NKEY, n] state codes, [NKEY, n] ei), positional: [NKEY] flags; nord: how many
    samples the kernel names (the request log is per sample of the kernel)."""
    src = jit.translate(func, pclass, fieldset, var_slot, field_ids)
    body = "\n".join("            " + ln for ln in src.case_body().split("\n"))
    code = SHIM % {"decl": "\n".join("    " + d for d in src.decl) or "    char unused;", "body": body,
                   "ordinal": "stage" if src.counter is None else f"{src.counter} - 1"}
    cpp = tmp_path / f"{func.__name__}.cpp"
    so = tmp_path / f"{func.__name__}.so"
    cpp.write_text(code)
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-w", str(cpp), "-o", str(so)], check=True)
    lib = C.CDLL(str(so))
    n = len(data["t"])
    pf = data["x"].dtype == np.float32
    cols = {k: np.array(data[k], dtype=np.float64) for k in ("t", "z", "y", "x", "dz", "dy", "dx", "dt")}  # (copies: data stays the input)
    state, pid = data["state"].copy(), data["particle_id"].copy()
    extras = [None] * 4
    for name, (slot, _) in var_slot.items():
        extras[slot] = data[name].copy()
    sam = np.ascontiguousarray(samples, dtype=np.float64) if samples is not None else np.zeros((NKEY, 3, n))
    assert sam.shape == (NKEY, 3, n)
    ns = C.c_int32(0)
    nsam = max(int(nord), 1)
    sstate, sei = effects if effects is not None else (np.zeros((NKEY, n), np.int32), np.zeros((NKEY, n), np.int32))
    sstate, sei = np.ascontiguousarray(sstate, dtype=np.int32), np.ascontiguousarray(sei, dtype=np.int32)
    ei0 = np.array(data["ei"][:, 0], dtype=np.int32)
    req = np.zeros((nsam, 6, n))
    asked = np.zeros((nsam, n), np.uint8)
    positional = np.ascontiguousarray(positional if positional is not None else np.zeros(NKEY), dtype=np.int32)
    assert sstate.shape == sei.shape == (NKEY, n) and positional.shape == (NKEY,)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
    lib.run(C.c_int64(n), C.c_int(int(pf)), *[ptr(cols[k]) for k in ("t", "z", "y", "x", "dz", "dy", "dx", "dt")], ptr(state), ptr(pid),
            *[ptr(e) for e in extras], ptr(sam), C.byref(ns), ptr(sstate), ptr(sei), ptr(ei0), ptr(req), ptr(asked), ptr(positional))
    src.requests, src.asked = req, asked.astype(bool)
    out = {k: cols[k].astype(data[k].dtype) for k in cols}
    out["state"] = state
    out["ei"] = ei0.reshape(-1, 1)
    for name, (slot, _) in var_slot.items():
        out[name] = extras[slot]
    return out, ns.value, src


def _check(func, tmp_path, *, spatial=np.float32, context=None, fields=None, n=400, seed=0, finite=False, grid=None, codes=(51, 60, 61, 70), positional=(), check_nsamples=True, check_log=True):
    P = pa.get_default_particle(spatial).add_variable([
        pa.Variable("age", dtype=np.float32, initial=0), pa.Variable("acc", dtype=np.float64, initial=0),
        pa.Variable("count", dtype=np.int32, initial=0), pa.Variable("flag", dtype=np.int64, initial=0)])
    data = _columns(P, n, seed, finite)
    var_slot = {"age": (0, "f32"), "acc": (1, "f64"), "count": (2, "i32"), "flag": (3, "i64")}
    rng = np.random.default_rng(seed + 100)
    # fields: {name: ncomp} (each sampled once, in this order) or [(name, ncomp), ...] = the samples in the order the kernel takes them
    order = list(fields.items()) if isinstance(fields, dict) else list(fields or [])
    fake_fields, field_ids = {}, {}
    log = []
    effects = (np.where(rng.random((NKEY, n)) < 0.1, rng.choice(list(codes), size=(NKEY, n)), 0).astype(np.int32),
               rng.integers(0, 1000, size=(NKEY, n)).astype(np.int32))
    nscalar = 0
    for name, ncomp in order:
        if name in fake_fields:
            continue
        if ncomp == 1:
            field_ids[name] = nscalar
            nscalar += 1
        key = 2 * 16 + field_ids[name] if ncomp == 1 else (0 if ncomp == 2 else 16)
        f = _FakeField([rng.normal(size=n) for _ in range(ncomp)], log, effects, positional=name in positional, key=key)
        f.name = name
        if grid is not None:
            f.grid = grid
        if ncomp > 1:
            f.U = f.V = None  # what marks a VectorField for the translator
        fake_fields[name] = f
    fs = _FakeFieldSet(context or {}, fake_fields)
    sam = np.zeros((NKEY, 3, n))
    pos = np.zeros(NKEY, np.int32)
    for name, f in fake_fields.items():
        for j, comp in enumerate(f.values):
            sam[f.key, j] = comp
        pos[f.key] = int(name in positional)
    fields = order
    got, nsamples, src = _run_translated(func, P, fs, data, var_slot, field_ids, sam, tmp_path, effects, pos, nord=len(order))
    assert nsamples == len(fields) or not check_nsamples  # (a conditional sample nobody takes is not counted)
    ref = {k: v.copy() for k, v in data.items()}
    with np.errstate(all="ignore"):
        func(HostParticles(ref, np.arange(n)), fs)
    for k in got:
        assert got[k].dtype == ref[k].dtype, k
        assert np.array_equal(got[k], ref[k], equal_nan=True), (func.__name__, k, np.flatnonzero(~((got[k] == ref[k]) | (np.isnan(got[k].astype(float)) & np.isnan(ref[k].astype(float)))))[:5])
    # every sample NumPy took: same field, same particles, same point (bit for bit), same float32-ness of y.  The k-th sample the kernel
    # names is the k-th entry of NumPy's log unless a guard skipped it there (`if np.any(mask): ...` with an empty selection, an early
    # return) -- then no lane of the translated kernel asked for it either
    assert len(log) == len(fields) or not check_log
    assert nsamples <= len(fields)
    j = 0
    for k, (name, ncomp) in enumerate(fields):
        lanes = np.flatnonzero(src.asked[k])
        entry = log[j] if j < len(log) else None
        if entry is not None and entry["field"] == name and (len(lanes) > 0 or np.size(entry["x"]) == 0):
            j += 1
        else:
            assert len(lanes) == 0, (func.__name__, "a sample NumPy did not take", k)
            continue
        rows = entry["rows"]
        if rows.ndim == 0:  # (a positional fake sampled without particles: the points must be those of the lanes that asked, in order)
            rows = lanes
            assert len(rows) == len(entry["x"])
            entry["rows"] = rows
        assert np.array_equal(lanes, rows), (func.__name__, "which particles take sample", k)
        for jj, c in enumerate("tzyx"):
            assert np.array_equal(src.requests[k, jj][rows], np.broadcast_to(entry[c], rows.shape), equal_nan=True), (func.__name__, "sample", k, c)
        assert np.all(src.requests[k, 4][rows] == float(entry["f32"])), (func.__name__, k)
        kind = {1: 2, 2: 0, 3: 1}[ncomp]
        assert np.all(src.requests[k, 5][rows] == kind * 100 + (field_ids[name] if ncomp == 1 else 0))
    assert j == len(log), (func.__name__, "samples NumPy took that the translated kernel did not", j, len(log))
    src.log = log
    return src


# ---- kernels --------------------------------------------------------------------------------------------------------------------------
def InPlace(particles, fieldset):
    particles.age += particles.dt          # f32 += f64
    particles.acc -= particles.age         # f64 -= f32
    particles.age *= 1.5                   # f32 *= Python float
    particles.acc /= particles.count + 10  # f64 /= int32
    particles.count += 2
    particles.flag -= particles.count      # int64 -= int32
    particles.dx += 0.1
    particles.dy *= particles.dt / 1200


def Promotion(particles, fieldset):
    a = particles.x + particles.age        # spatial dtype + f32
    b = particles.count * particles.age    # int32 * f32 -> f64
    c = particles.count / 4                # true division
    d = particles.count * 3 + particles.flag
    particles.acc = a + b - c + d / 2 + particles.x * particles.x
    particles.age = particles.acc          # f64 into f32
    particles.count = particles.age * 0.7  # float into int32: truncation (NaN / huge values: undefined, masked out below)
    particles.flag = np.where(particles.count > 2, particles.flag, 11)


def Functions(particles, fieldset):
    particles.acc = np.fmod(particles.acc, 2.5) + np.abs(particles.age) % 1.25 + (-particles.age) % fieldset.m
    particles.age = np.minimum(np.maximum(particles.age, -1), particles.acc) + np.sqrt(np.abs(particles.x)) ** 2
    particles.dz = np.clip(particles.dz, -0.5, fieldset.hi) + np.floor(particles.dy) - np.ceil(particles.dx)
    particles.count = np.abs(particles.count) - np.where(np.isnan(particles.acc), 1, 0)
    particles.flag = np.where(np.isfinite(particles.age) & (particles.flag != 3), particles.flag, -particles.flag)


def Masks(particles, fieldset):
    err = particles.state >= 50
    particles[err].state = StatusCode.Delete
    m = (particles.x > 0.5) & ~(particles.y < -1) | (particles.count == 0)
    particles.dx[m] = 0
    particles[np.logical_and(m, particles.age > 1)].acc = fieldset.k * 2
    particles.age[particles.flag < 0] += 3
    particles.state = np.where(np.logical_not(err) & (particles.acc > 4), StatusCode.StopExecution, particles.state)
    particles.dt = np.where(particles.t >= 1.0, 300.0, particles.dt)


def Samples(particles, fieldset):
    tval = fieldset.T[particles]
    particles.age = tval * 2 - particles.age
    u, v = fieldset.UV[particles]
    particles.acc += np.sqrt(u**2 + v**2)
    _, particles.dy, w = fieldset.UVW[particles]
    particles.dz = w + fieldset.T2[particles]


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_in_place_operators(tmp_path, spatial):
    _check(InPlace, tmp_path, spatial=spatial)


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_promotion_and_casts(tmp_path, spatial):
    # float -> int32 of NaN / out-of-range values is undefined in C and in NumPy alike: finite columns here
    _check(Promotion, tmp_path, spatial=spatial, seed=1, finite=True)


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_numpy_functions(tmp_path, spatial):
    _check(Functions, tmp_path, spatial=spatial, context={"m": 0.75, "hi": np.float32(0.25)}, seed=2)


def test_masks_and_states(tmp_path):
    src = _check(Masks, tmp_path, context={"k": 1.25}, seed=3)
    assert len(src.stages) == 1 and "state" in src.touched


def test_field_samples_are_stage_boundaries(tmp_path):
    src = _check(Samples, tmp_path, spatial=np.float64, fields={"T": 1, "UV": 2, "UVW": 3, "T2": 1}, seed=4)
    assert len(src.stages) == 5  # four samples: five stages


def RK2AtPoints(particles, fieldset):
    """The tutorials' hand-written mid-point scheme: the second sample is taken at a computed point, with the particles attached."""
    u1, v1 = fieldset.UV[particles]
    x1 = particles.x + u1 * 0.5 * particles.dt          # spatial dtype + f64 -> f64
    y1 = particles.y + v1 * 0.5 * particles.dt
    u2, v2 = fieldset.UV[particles.t + 0.5 * particles.dt, particles.z, y1, x1, particles]
    particles.dx += u2 * particles.dt
    particles.dy += v2 * particles.dt
    particles.acc = fieldset.T[particles.t, particles.z, particles.y, particles.x + 0.01, particles]   # y stays the stored dtype: f32 flag


def DetachedSamples(particles, fieldset):
    """`fieldset.F[t, z, y, x]` without the particles: the value only -- state and `ei` are left alone (field.py:173-176)."""
    particles.acc = fieldset.T[particles.t, particles.z, particles.y, particles.x]
    east = fieldset.T[particles.t, particles.z, particles.y, particles.x + fieldset.h]
    west = fieldset.T[particles.t + particles.dt, particles.z * 1.0, particles.y - particles.age, particles.x - fieldset.h]
    particles.age = (east - west) / (2 * fieldset.h)
    u, v = fieldset.UV[particles.t, particles.z, particles.y, particles.x]
    particles.dz = u + v + fieldset.T2[particles]


class _RectGrid:
    is_curvilinear = False


class _CurvGrid:
    is_curvilinear = True


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_samples_at_computed_points(tmp_path, spatial):
    src = _check(RK2AtPoints, tmp_path, spatial=spatial, fields=[("UV", 2), ("UV", 2), ("T", 1)], seed=21, finite=True)
    assert not src.detached and [e["attached"] for e in src.log] == [True, True, True]
    assert [bool(e["f32"]) for e in src.log] == [spatial == np.float32, False, spatial == np.float32]


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_samples_without_the_particles_leave_state_and_ei_alone(tmp_path, spatial):
    src = _check(DetachedSamples, tmp_path, spatial=spatial, context={"h": 0.25}, fields=[("T", 1), ("T", 1), ("T", 1), ("UV", 2), ("T2", 1)],
                 seed=22, finite=True, grid=_RectGrid(), codes=(51, 60, 61))
    assert src.detached and [e["attached"] for e in src.log] == [False, False, False, False, True]


def BareSample(particles, fieldset):
    fieldset.UV[particles.t + 400 * 86400, particles.z, particles.y, particles.x, particles]   # (the reference's FieldAccessOutsideTime)
    particles.acc = 1


def test_a_sample_for_its_effect_on_the_state(tmp_path):
    src = _check(BareSample, tmp_path, fields=[("UV", 2)], seed=23, finite=True)
    assert len(src.stages) == 2


def test_sample_without_particles_needs_a_rectilinear_grid(tmp_path):
    P = pa.get_default_particle(np.float32).add_variable([pa.Variable("acc", dtype=np.float64, initial=0)])

    def K(particles, fieldset):
        particles.acc = fieldset.T[particles.t, particles.z, particles.y, particles.x]

    f = _FakeField([np.zeros(3)])
    f.grid = _CurvGrid()
    with pytest.raises(jit.NotTranslatable, match="curvilinear"):
        jit.translate(K, P, _FakeFieldSet({}, {"T": f}), {"acc": (0, "f64")}, {"T": 0})
    f2 = _FakeField([np.zeros(3)])
    with pytest.raises(jit.NotTranslatable, match="not known"):
        jit.translate(K, P, _FakeFieldSet({}, {"T": f2}), {"acc": (0, "f64")}, {"T": 0})

    def Scalar(particles, fieldset):
        particles.acc = fieldset.T[particles.t, 0.5, particles.y, particles.x, particles]

    with pytest.raises(jit.NotTranslatable, match="not a numeric array"):
        jit.translate(Scalar, P, _FakeFieldSet({}, {"T": f}), {"acc": (0, "f64")}, {"T": 0})


def Trig(particles, fieldset):
    particles.acc = np.deg2rad(particles.y) * 2 + np.rad2deg(particles.age)
    particles.age = np.deg2rad(particles.age)


def Libm(particles, fieldset):
    lat = np.deg2rad(particles.y)
    particles.acc = np.cos(lat) * np.sin(particles.x) + np.exp(-np.abs(particles.age)) + np.arctan2(particles.dy, particles.dx)
    particles.age = np.cos(particles.age) + np.hypot(particles.age, 2)


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_degrees_and_radians_are_exact(tmp_path, spatial):
    _check(Trig, tmp_path, spatial=spatial, seed=5)


def test_transcendental_functions_only_on_request(tmp_path, monkeypatch):
    """np.sin & co are within an ulp or so of NumPy's on the device, not bit-identical: the translator takes them only under
    PARCELS_AMD_JIT_LIBM=1 (here the host's libm stands in for the device's: same order of agreement)."""
    P = pa.get_default_particle(np.float64).add_variable([pa.Variable("age", dtype=np.float32, initial=0), pa.Variable("acc", dtype=np.float64, initial=0)])
    fs = _FakeFieldSet({}, {})
    monkeypatch.delenv("PARCELS_AMD_JIT_LIBM", raising=False)
    with pytest.raises(jit.NotTranslatable, match="PARCELS_AMD_JIT_LIBM"):
        jit.translate(Libm, P, fs, {"age": (0, "f32"), "acc": (1, "f64")}, {})
    monkeypatch.setenv("PARCELS_AMD_JIT_LIBM", "1")
    n = 300
    Pfull = pa.get_default_particle(np.float64).add_variable([
        pa.Variable("age", dtype=np.float32, initial=0), pa.Variable("acc", dtype=np.float64, initial=0),
        pa.Variable("count", dtype=np.int32, initial=0), pa.Variable("flag", dtype=np.int64, initial=0)])
    data = _columns(Pfull, n, 6, finite=True)
    var_slot = {"age": (0, "f32"), "acc": (1, "f64"), "count": (2, "i32"), "flag": (3, "i64")}
    got, _, _ = _run_translated(Libm, Pfull, fs, data, var_slot, {}, None, tmp_path)
    ref = {k: v.copy() for k, v in data.items()}
    Libm(HostParticles(ref, np.arange(n)), fs)
    # (`np.exp(-np.abs(particles.age))` is a FLOAT32 exponential of the float32 Variable: its ulp is 1.2e-7)
    np.testing.assert_allclose(got["acc"], ref["acc"], rtol=5e-7, atol=5e-7)
    np.testing.assert_allclose(got["age"], ref["age"], rtol=5e-7)


def ByName(particles, fieldset):
    a = np.add(particles.age, particles.count)            # f32 + int32 -> f64
    b = np.multiply(np.subtract(particles.x, 0.5), np.negative(particles.age))
    particles.acc = np.divide(a, 3) + b - np.mod(particles.acc, 2) + np.remainder(particles.age, -1.5)
    particles.age = np.square(particles.age) - particles.dy**2 + np.sign(particles.dx) * np.sign(particles.acc)
    particles.flag = np.sign(particles.flag) * 4 + np.where(np.less_equal(particles.count, 2) | np.not_equal(particles.state, 0), 1, 0)
    particles.count = np.where(np.greater(particles.acc, particles.age) & np.equal(particles.count, particles.count), 5, particles.count)


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_ufuncs_called_by_name(tmp_path, spatial):
    _check(ByName, tmp_path, spatial=spatial, seed=11)


def Branches(particles, fieldset):
    if fieldset.mode == 1:
        particles.acc += 1
    elif fieldset.mode == 2:
        particles.acc += 2
        if STRICT:
            particles.age = 0
    else:
        particles.acc -= 1
    tmp = particles.x * 2          # a temporary: in-place operators keep its dtype
    tmp += particles.dt            # spatial dtype += f64 -> cast back
    tmp /= 3
    cnt = particles.count + 1
    cnt *= 2
    particles.dz = tmp
    particles.flag = cnt


STRICT = True


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_constant_branches_and_local_temporaries(tmp_path, mode, spatial):
    src = _check(Branches, tmp_path, spatial=spatial, context={"mode": mode}, seed=7 + mode)
    assert ("age" in src.touched) == (mode == 2)


def test_what_the_translator_refuses():
    P = pa.get_default_particle(np.float32).add_variable(pa.Variable("age", dtype=np.float32, initial=0))
    fs = _FakeFieldSet({}, {})

    def k_if(particles, fieldset):
        if particles.age.max() > 1:
            particles.age += 1

    def k_reduce(particles, fieldset):
        particles.age += len(particles)

    def k_random(particles, fieldset):
        particles.age = np.random.rand(len(particles))

    def k_trig(particles, fieldset):
        particles.age = np.sin(particles.x)

    def k_intcast(particles, fieldset):
        particles.state += 0.5  # NumPy raises (same_kind): the host path raises it for the user

    def k_unknown(particles, fieldset):
        particles.nope = 1

    def k_view_inplace(particles, fieldset):
        a = particles.age
        a += 1

    for f, word in ((k_if, "`if` on something other"), (k_view_inplace, "particle column's view"), (k_reduce, "call"), (k_random, "call"), (k_trig, "np.sin"), (k_intcast, "does not cast back"),
                    (k_unknown, "no Variable")):
        with pytest.raises(jit.NotTranslatable, match=word):
            jit.translate(f, P, fs, {"age": (0, "f32")}, {})
    assert jit.candidate_variables(k_unknown, P) == [] and jit.candidate_variables(k_trig, P) == ["age"]


def test_ranks_of_one_node_can_compile_the_same_module_at_once(tmp_path):
    """One process per GPU: every rank translates the same kernel list and builds the same module into the same cache directory at the same
    time.  Private temporaries + an atomic rename: every build succeeds, one module remains, and it exports the launcher."""
    script = tmp_path / "build_one.py"
    script.write_text(f"""
import sys
sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})
sys.path.insert(0, {repr(os.path.dirname(os.path.abspath(__file__)))})
import numpy as np
import parcels_amd as pa
from parcels_amd import jit
import test_jit_translator as T
P = pa.get_default_particle(np.float32).add_variable(pa.Variable("age", dtype=np.float32, initial=0))
src = jit.translate(T.InPlaceAge, P, T._FakeFieldSet({{}}, {{}}), {{"age": (0, "f32")}}, {{}}, slot_prefix="k0_")
prog = jit.UserProgram([src], 0, 1, fast=1, particles_f32=True)
print(prog.build())
""")
    env = dict(os.environ, PARCELS_AMD_JIT_CACHE=str(tmp_path / "cache"))
    procs = [subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(4)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    paths = {o[0].strip().split("\\n")[-1] for o in outs}
    assert len(paths) == 1
    files = sorted(os.listdir(tmp_path / "cache"))
    assert [f for f in files if f.endswith(".so")] == [os.path.basename(paths.pop())] and not [f for f in files if f.endswith(".tmp")]
    sym = subprocess.run(["nm", "-D", str(tmp_path / "cache" / [f for f in files if f.endswith(".so")][0])], capture_output=True, text=True).stdout
    assert " T pk_user_launch" in sym


def InPlaceAge(particles, fieldset):
    particles.age += particles.dt


# ---- selections of the particles (ParticleSetView of the reference: particlesetview.py) --------------------------------------------------
OUT_OF_BOUNDS_STATES = [StatusCode.ErrorOutOfBounds, StatusCode.ErrorThroughSurface]
SPEED, DRIFT_DEPTH, MAX_DEPTH, MIN_DEPTH, DRIFT_TIME, CYCLE_TIME = 0.1, 2.0, 4.0, -1.0, 1800.0, 3000.0


def ArgoLike(particles, fieldset):
    """tutorial_Argofloats.ipynb: a state machine over sub-selections of the particles bound to locals, masks within them, a sample
    at the positions of one of them."""
    ptcls0 = particles[particles.count == 0]
    ptcls1 = particles[particles.count == 1]
    ptcls2 = particles[particles.count == 2]
    ptcls3 = particles[particles.count == 3]
    ptcls4 = particles[particles.count == 4]
    ptcls0.dz += SPEED * ptcls0.dt
    next_phase = ptcls0.z + ptcls0.dz >= DRIFT_DEPTH
    ptcls0.count[next_phase] = 1
    ptcls0.dz[next_phase] = DRIFT_DEPTH - ptcls0.z[next_phase]
    ptcls1.age += ptcls1.dt
    next_phase = ptcls1.age >= DRIFT_TIME
    ptcls1.count[next_phase] = 2
    ptcls1.age[next_phase] = 0
    ptcls2.dz += SPEED * ptcls2.dt
    next_phase = ptcls2.z + ptcls2.dz >= MAX_DEPTH
    ptcls2.count[next_phase] = 3
    ptcls2.dz[next_phase] = MAX_DEPTH - ptcls2.z[next_phase]
    ptcls3.dz -= SPEED * ptcls3.dt
    ptcls3.acc = fieldset.T[ptcls3.t, ptcls3.z, ptcls3.y, ptcls3.x]
    next_phase = ptcls3.z + ptcls3.dz <= MIN_DEPTH
    ptcls3.count[next_phase] = 4
    ptcls3.dz[next_phase] = MIN_DEPTH - ptcls3.z[next_phase]
    next_phase = ptcls4.acc >= CYCLE_TIME
    ptcls4.count[next_phase] = 0
    ptcls4.acc[next_phase] = 0
    ptcls4.age = np.nan
    particles.acc += particles.dt


def NearShore(particles, fieldset):
    """tutorial_unstuck_Agrid.ipynb: a second sample for the particles the first one selects."""
    particles.age = fieldset.T[particles]
    near = particles[particles.age < 0.5]
    dU, dV = fieldset.UV[near]
    near.dx += dU * near.dt
    near.dy += dV * near.dt
    particles.acc = fieldset.T2[particles.t, particles.z, particles.y, particles.x, particles]


def Recovery(particles, fieldset):
    """tutorial_statuscodes.md / tutorial_schism.ipynb / tests: recovery kernels over selections."""
    through_surface = particles.state == StatusCode.ErrorThroughSurface
    particles[through_surface].dz = fieldset.surface - particles[through_surface].z
    particles[through_surface].state = StatusCode.Evaluate
    oob = np.isin(particles.state, OUT_OF_BOUNDS_STATES)
    particles[oob].flag = 1
    particles[oob].state = StatusCode.Delete
    inds = np.where(particles.state == StatusCode.ErrorOutsideTimeInterval)
    particles[inds].dx -= 1.0
    particles[inds].state = StatusCode.StopExecution
    rows = np.argwhere(particles.state == StatusCode.Success).flatten()
    u, v = fieldset.UV[particles[rows]]
    particles[rows].dx = u * particles[rows].dt
    particles[rows].dy = v * particles[rows].dt
    particles[rows].state = StatusCode.Evaluate


def Heating(particles, fieldset):
    """tests/test_particlesetview.py: in-place operators through inline selections, groups chosen by earlier stores."""
    particles[particles.x < 0.5].age += 1.0
    particles[particles.x >= 0.5].age -= 0.5
    particles[particles.x < -0.33].count = 1
    particles[(particles.x >= -0.33) & (particles.x < 0.67)].count = 2
    particles[particles.x >= 0.67].count = 3
    particles[particles.count == 1].flag += 1
    particles[particles.count == 2].flag += 2
    sel = particles[particles.count == 3]
    sel[sel.flag < 0].acc = sel[sel.flag < 0].acc * 2 + sel[sel.flag < 0].x


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_state_machine_over_selections(tmp_path, spatial):
    src = _check(ArgoLike, tmp_path, spatial=spatial, fields=[("T", 1)], seed=31, finite=True, grid=_RectGrid(), codes=(51, 60, 61), positional=("T",))
    assert src.counter is not None and src.detached and 0 < len(src.log[0]["rows"]) < 400  # only the particles in phase 3 take the sample


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_a_sample_for_the_particles_an_earlier_sample_selects(tmp_path, spatial):
    src = _check(NearShore, tmp_path, spatial=spatial, fields=[("T", 1), ("UV", 2), ("T2", 1)], seed=32, finite=True)
    assert src.counter is not None and 0 < len(src.log[1]["rows"]) < 400 and len(src.log[2]["rows"]) == 400


def test_recovery_kernels_over_selections(tmp_path):
    src = _check(Recovery, tmp_path, context={"surface": 0.25}, fields=[("UV", 2)], seed=33, finite=True)
    assert 0 < len(src.log[0]["rows"]) < 400


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_inline_selections(tmp_path, spatial):
    _check(Heating, tmp_path, spatial=spatial, seed=34, finite=True)


def test_arrays_over_different_selections_do_not_mix(tmp_path):
    P = pa.get_default_particle(np.float32).add_variable([pa.Variable("acc", dtype=np.float64, initial=0)])
    fs = _FakeFieldSet({}, {})

    def Mixed(particles, fieldset):
        a = particles[particles.x > 0]
        particles.acc = a.x * 2            # (NumPy: shapes differ)

    def Mixed2(particles, fieldset):
        a = particles[particles.x > 0]
        b = particles[particles.y > 0]
        a.acc = a.x + b.x

    def AsValue(particles, fieldset):
        a = particles[particles.x > 0]
        particles.acc = a

    for k, pat in ((Mixed, "another selection"), (Mixed2, "different selections"), (AsValue, "used as a value")):
        with pytest.raises(jit.NotTranslatable, match=pat):
            jit.translate(k, P, fs, {"acc": (0, "f64")}, {})


def test_kernels_that_set_success_stay_on_the_host_path():
    """kernel.py:190-193: a particle in state Success is evaluated for as long as ANY particle is in state Evaluate -- a property of the whole
    set (tests/test_particleset_execute.py's MoveLeft relies on it); so are states the translator cannot bound."""
    P = pa.get_default_particle(np.float32).add_variable([pa.Variable("count", dtype=np.int32, initial=0)])
    fs = _FakeFieldSet({}, {})

    def MoveLeft(particles, fieldset):
        inds = np.where(particles.state == StatusCode.ErrorOutOfBounds)
        particles[inds].dx -= 1.0
        particles[inds].state = StatusCode.Success

    def ViaWhere(particles, fieldset):
        particles.state = np.where(particles.x > 0, 0, particles.state)

    def Computed(particles, fieldset):
        particles.state = particles.count + 10

    def Fine(particles, fieldset):
        particles.state = np.where(particles.x > 0, StatusCode.Delete, np.where(particles.y > 0, StatusCode.StopExecution, particles.state))
        particles[particles.state == StatusCode.ErrorOutOfBounds].state = StatusCode.Evaluate

    for k, pat in ((MoveLeft, "StatusCode.Success"), (ViaWhere, "StatusCode.Success"), (Computed, "computed value")):
        with pytest.raises(jit.NotTranslatable, match=pat):
            jit.translate(k, P, fs, {"count": (0, "i32")}, {})
    jit.translate(Fine, P, fs, {"count": (0, "i32")}, {})


def StateAroundSample(particles, fieldset):
    before = particles.state * 1
    particles.acc = particles.state + fieldset.T[particles]      # the state as it was BEFORE the sample marked the particles it fails on
    particles.flag = particles.state - before                    # ... and after
    particles.count = np.where(particles.state >= 50, 1, 0) + fieldset.T2[particles] * 0


def test_state_is_read_where_the_statement_reads_it(tmp_path):
    """A sample may change particles.state (field.py:307-378); an expression that read the state before the sample keeps that value."""
    src = _check(StateAroundSample, tmp_path, fields=[("T", 1), ("T2", 1)], seed=41, finite=True)
    assert len(src.stages) == 3


def test_a_selection_written_twice_around_a_sample_is_not_one_selection():
    """Python evaluates the value first -- the sample for the inline selection -- and the target's selection afterwards, when states may have
    changed: NumPy would need equal shapes.  (A selection bound to a name is evaluated once, and fine.)"""
    P = pa.get_default_particle(np.float32).add_variable([pa.Variable("age", dtype=np.float32, initial=0)])
    f = _FakeField([np.zeros(3)])

    def Inline(particles, fieldset):
        particles[particles.state == 10].age = fieldset.T[particles[particles.state == 10]]

    def Named(particles, fieldset):
        ok = particles[particles.state == 10]
        ok.age = fieldset.T[ok]

    with pytest.raises(jit.NotTranslatable, match="another selection"):
        jit.translate(Inline, P, _FakeFieldSet({}, {"T": f}), {"age": (0, "f32")}, {"T": 0})
    assert jit.translate(Named, P, _FakeFieldSet({}, {"T": f}), {"age": (0, "f32")}, {"T": 0}).counter is not None


def SubmergeAsWritten(particles, fieldset):
    """tests/test_advection.py's SubmergeParticle, guards included: whether a selection is empty is a property of the whole set, but
    everything behind the guard only touches that selection -- it does nothing when the selection is empty, so the guard can go."""
    if len(particles.state) == 0:
        return
    inds = np.argwhere(particles.state == fieldset.code).flatten()
    if len(inds) == 0:
        return
    u, v = fieldset.UV[particles[inds]]
    particles[inds].dx = u * particles[inds].dt
    particles[inds].dy = v * particles[inds].dt
    particles[inds].dz = 0.0
    particles[inds].z = 0
    particles[inds].state = StatusCode.Evaluate


def GuardedBlocks(particles, fieldset):
    particles.acc += 1
    hot = particles.age > fieldset.limit
    if np.any(hot):
        particles[hot].age = 0
        particles[hot].count += 1
    sel = particles[particles.flag < fieldset.code]
    if len(sel) > 0:
        sel.flag = fieldset.T[sel] * 0 + 7
    if not len(particles):
        return
    particles.dz = 0.25


@pytest.mark.parametrize("code", [61, 52])   # 52: no particle is in that state -- the selection is empty and the kernel returns early
def test_emptiness_guards_around_code_confined_to_the_selection(tmp_path, code):
    src = _check(SubmergeAsWritten, tmp_path, context={"code": code}, fields=[("UV", 2)], seed=51, finite=True, check_nsamples=False, check_log=False)
    assert (len(src.log) == 0) == (code == 52)
    _check(GuardedBlocks, tmp_path, context={"limit": 1.0 if code == 61 else 1e9, "code": 3 if code == 61 else -100}, fields=[("T", 1)], seed=52,
           finite=True, check_nsamples=False, check_log=False)


def test_an_emptiness_test_decides_for_the_whole_set_when_the_code_behind_it_leaves_the_selection():
    P = pa.get_default_particle(np.float32).add_variable([pa.Variable("acc", dtype=np.float64, initial=0)])
    f = _FakeField([np.zeros(3)])
    fs = _FakeFieldSet({}, {"T": f})

    def LeavesIt(particles, fieldset):
        inds = np.where(particles.state == 61)
        if len(inds) == 0:   # (np.where's tuple has length 1: never empty -- but that is for NumPy to say)
            return
        particles.acc += 1

    def LeavesIt2(particles, fieldset):
        m = particles.x > 0
        if np.any(m):
            particles.acc = fieldset.T[particles]

    def ElseBranch(particles, fieldset):
        m = particles.x > 0
        if np.any(m):
            particles[m].acc = 1
        else:
            particles.acc = 2

    for k in (LeavesIt, LeavesIt2, ElseBranch):
        with pytest.raises(jit.NotTranslatable, match="outside the selection|`if` on something other"):
            jit.translate(k, P, fs, {"acc": (0, "f64")}, {"T": 0})


def LocalArrays(particles, fieldset):
    """Temporaries made by the array constructors and filled by masked item assignment (tutorial_nestedgrids.ipynb's u / v buffers)."""
    u = np.zeros_like(particles.x)            # the spatial dtype
    v = np.zeros(particles.x.shape)           # float64
    w = np.full_like(particles.count, 7)      # int32
    k = np.ones(len(particles.dt))
    east = particles.x > 0
    u[east] = particles.age[east] * 2         # float32 values into a spatial-dtype temporary
    v[particles.y > 0] = 1.5
    v[east] += particles.acc[east]
    w[particles.flag < 0] -= 3
    k[np.where(particles.count > 2)] *= 0.5
    particles.dx += u * particles.dt
    particles.acc = v + k + w
    near = particles[particles.dy < 0]
    buf = np.zeros_like(near.acc)             # on the selection
    buf[near.count > 0] = near.dt[near.count > 0]
    near.acc += buf
    full = np.full(particles.age.shape, fieldset.c)
    scratch = np.empty_like(particles.age)     # (contents undefined until written)
    scratch[east] = 1
    scratch[~east] = full[~east]
    particles.age = scratch


def SharedTemporary(particles, fieldset):
    u = particles.x * 2
    w = u
    w[particles.y > 0] = 0     # changes u as well: one ndarray under two names
    particles.acc = u


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_array_constructors_and_item_assignment_on_temporaries(tmp_path, spatial):
    _check(LocalArrays, tmp_path, spatial=spatial, context={"c": 0.25}, seed=61, finite=True)


def test_one_array_under_two_names_is_left_to_numpy():
    P = pa.get_default_particle(np.float32).add_variable([pa.Variable("acc", dtype=np.float64, initial=0)])
    with pytest.raises(jit.NotTranslatable, match="two names"):
        jit.translate(SharedTemporary, P, _FakeFieldSet({}, {}), {"acc": (0, "f64")}, {})
