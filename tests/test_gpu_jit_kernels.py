"""GPU: user-written Python kernels compiled into the fused step loop (parcels_amd/jit.py) against the host path
(parcels_amd/hostkernels.py: the reference's loop of kernel.py:190-245 on the host columns, i.e. NumPy itself).  Every supported
construct -- in-place operators across dtypes, np.where, masked assignment, comparisons and boolean algebra, %, fmod, minimum / maximum,
scalar and vector field samples, changes of dt and of the state -- gives the same columns bit for bit either way, the compiled list is ONE
launch of the kernel-list interpreter, and whatever the translator does not take (control flow, random numbers, integer Variables ...)
still runs, on the host path."""
import os

import numpy as np
import pytest

import parcels_amd as pa
from parcels_amd import StatusCode

pytestmark = pytest.mark.gpu


def _fieldset(mesh="flat", seed=4):
    """U, V and a scalar T on a small rectilinear A-grid with three time levels (smooth random data)."""
    nx, ny, nt = 24, 18, 3
    md = pa.SGrid2DMetadata(node_dimensions=("XG", "YG"), node_coordinates=("lon", "lat"),
                            face_dimensions=(pa.FaceNodePadding("XC", "XG", pa.Padding.LOW), pa.FaceNodePadding("YC", "YG", pa.Padding.LOW)),
                            vertical_dimensions=None)
    if mesh == "flat":
        lon, lat, vel = np.linspace(0.0, 4.0e4, nx), np.linspace(0.0, 3.0e4, ny), 0.1
    else:
        lon, lat, vel = np.linspace(-20.0, 20.0, nx), np.linspace(-15.0, 15.0, ny), 1.5
    coords = {"lon": (("XG",), lon), "lat": (("YG",), lat), "time": (("time",), np.arange(nt) * 43200.0)}
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.linspace(0, 1, ny), np.linspace(0, 1, nx), indexing="ij")

    def smooth(scale):
        a = sum(rng.normal() * np.sin(2 * np.pi * (kx * xx + ky * yy) + rng.uniform(0, 6)) for kx in (1, 2) for ky in (1, 2))
        return np.stack([scale * a * (1 + 0.1 * k) for k in range(nt)])

    dims = ("time", "YG", "XG")
    data = {"U": (dims, smooth(vel)), "V": (dims, smooth(vel)), "T": (dims, 10.0 + smooth(3.0)), "S": (dims, smooth(1.0).astype(np.float32))}
    return pa.FieldSet.from_sgrid_conventions(pa.Dataset(data, coords, sgrid=md), mesh=mesh)


def _pclass(spatial=np.float32):
    P = pa.get_default_particle(spatial)
    return P.add_variable([pa.Variable("age", dtype=np.float32, initial=0), pa.Variable("acc", dtype=np.float64, initial=0),
                           pa.Variable("temp", dtype=np.float32, initial=0), pa.Variable("speed", dtype=np.float64, initial=0),
                           pa.Variable("count", dtype=np.int32, initial=0), pa.Variable("flag", dtype=np.int64, initial=-1)])


def _run(kernels, *, jit, mesh="flat", spatial=np.float32, n=300, runtime=12 * 600.0, dt=600.0, context=None, margin=0.15, expect_error=None,
         output=None, seed=1):
    old = os.environ.get("PARCELS_AMD_JIT")
    os.environ["PARCELS_AMD_JIT"] = "1" if jit else "0"
    try:
        fs = _fieldset(mesh)
        for k, v in (context or {}).items():
            fs.add_context(k, v)
        rng = np.random.default_rng(seed)
        lon, lat = np.asarray(fs.U.grid.lon), np.asarray(fs.U.grid.lat)
        x = lon[0] + (margin + (1 - 2 * margin) * rng.uniform(size=n)) * (lon[-1] - lon[0])
        y = lat[0] + (margin + (1 - 2 * margin) * rng.uniform(size=n)) * (lat[-1] - lat[0])
        pset = pa.ParticleSet(fs, pclass=_pclass(spatial), x=x, y=y, t=np.where(np.arange(n) % 4 == 0, 600.0, 0.0))
        err = None
        kw = {"output_file": output} if output is not None else {}
        try:
            pset.execute(kernels, runtime=runtime, dt=dt, **kw)
        except (pa.FieldOutOfBoundError, pa.FieldOutOfBoundSurfaceError) as e:
            err = type(e).__name__
        assert err == expect_error, err
        return pset, {k: np.array(v) for k, v in pset._data.items()}
    finally:
        if old is None:
            os.environ.pop("PARCELS_AMD_JIT", None)
        else:
            os.environ["PARCELS_AMD_JIT"] = old


def _both(kernels, **kw):
    """Run the list compiled and on the host path; return the compiled run after asserting both agree bit for bit."""
    pj, dj = _run(kernels, jit=True, **kw)
    ph, dh = _run(kernels, jit=False, **kw)
    assert pj._kernel.user_program is not None and not pj._kernel.host_functions, pj._kernel.jit_report
    assert pj._last_stats["launches"] >= 1 and not pj._last_stats.get("hosted"), pj._last_stats
    assert ph._kernel.user_program is None and (ph._last_stats is None or ph._last_stats.get("hosted")), ph._kernel.jit_report  # (None: it raised)
    assert set(dj) == set(dh)
    for k in dj:
        assert dj[k].dtype == dh[k].dtype and np.array_equal(dj[k], dh[k], equal_nan=True), (k, np.flatnonzero(dj[k] != dh[k])[:5], dj[k][:4], dh[k][:4])
    return pj, dj


# ---- the kernels (module level: the translator reads their source) ----------------------------------------------------------------
def Age(particles, fieldset):
    particles.age += particles.dt  # float32 += float64: computed in float64, stored as float32


def DeleteOld(particles, fieldset):
    particles.state = np.where(particles.age > fieldset.max_age, StatusCode.Delete, particles.state)


def DeleteErrorParticle(particles, fieldset):
    any_error = particles.state >= 50
    particles[any_error].state = StatusCode.Delete


def PeriodicBC(particles, fieldset):
    particles.acc += particles.dx
    particles.x = np.fmod(particles.x, fieldset.width)


def SampleT(particles, fieldset):
    particles.temp = fieldset.T[particles]


def SampleS32(particles, fieldset):
    particles.temp = fieldset.S[particles]


def SampleExpr(particles, fieldset):
    t = fieldset.T[particles]
    s = fieldset.S[particles]
    particles.acc = t * 2 + s / 3 - particles.age
    particles.temp += s


def SampleSpeed(particles, fieldset):
    u, v = fieldset.UV[particles]
    particles.speed = np.sqrt(u**2 + v**2)
    _, particles.acc = fieldset.UV[particles]


def Algebra(particles, fieldset):
    m = (particles.x > fieldset.x0) & ~(particles.y < fieldset.y0) | (particles.age == 0)
    particles.acc = np.where(m, particles.acc + 1, particles.acc - 0.25)
    particles.temp = np.minimum(np.maximum(particles.temp + particles.dx * 1000, -5), 7.5) + np.abs(particles.dy) % 0.3
    particles.age = np.floor(particles.age / 7) * 7 + 1
    particles.speed = particles.state / 3 + particles.particle_id * 2


def ChangeDt(particles, fieldset):
    particles.dt = np.where(particles.t >= 3000.0, 300.0, particles.dt)
    particles.dx[particles.age > 2000] = 0


def Kick(particles, fieldset):
    particles.dx += 0.1
    particles.dy -= fieldset.kick


def Counters(particles, fieldset):
    particles.count += 1                                        # int32 += Python int stays int32
    moved = np.logical_and(np.abs(particles.dx) > 0, ~np.isnan(particles.dy))
    particles.flag = np.where(moved, particles.count * 3 - particles.flag, particles.flag)  # int32 * int + int64 -> int64
    particles.count[particles.age > 3000] = 7.9                 # a float into an integer column truncates
    particles.acc = particles.count / 2 + np.clip(particles.flag, -2, 5)  # integer / integer is a float64 division


def NotElementwise(particles, fieldset):
    if len(particles) > 0:  # control flow + a reduction: the host path
        particles.acc += len(particles)


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_age_and_delete(gpu, spatial):
    p, d = _both([pa.AdvectionRK4, Age, DeleteOld], spatial=spatial, context={"max_age": 4000.0})
    assert 0 < len(p) < 300 or np.all(d["age"] <= 4000.0)  # particles older than max_age were deleted on the way
    # ONE launch -- of the dedicated A-grid kernel (pk_exec_stats.program 100), the two user kernels riding along: they sample no field
    assert p._last_stats["program"] == 100 and p._kernel.user_program.flags == 1


@pytest.mark.parametrize("mesh", ["flat", "spherical"])
def test_recovery_kernel_written_by_the_user(gpu, mesh):
    """tutorials: particles that leave the domain are deleted by a user kernel behind the advection kernel."""
    p, d = _both([pa.AdvectionRK4, DeleteErrorParticle], mesh=mesh, margin=0.01, runtime=40 * 600.0, n=400)
    assert len(p) < 400  # some did leave


def test_periodic_boundary_and_accumulator(gpu):
    _both([pa.AdvectionEE, PeriodicBC], context={"width": 3.0e4}, spatial=np.float32)
    _both([pa.AdvectionEE, PeriodicBC], context={"width": 3.0e4}, spatial=np.float64)


def test_scalar_and_vector_samples(gpu):
    p, _ = _both([pa.AdvectionRK4, SampleT])
    # T shares U's grid, dtype and layout: the sampling kernel rides in the dedicated A-grid kernel (FastA::S, eval_scalar_fast)
    assert p._last_stats["program"] == 100 and p._kernel.user_program.flags & 1 and p._kernel.user_program.sample_fids
    p, _ = _both([pa.AdvectionRK4, SampleS32])
    assert p._last_stats["program"] == 2 and not p._kernel.user_program.flags & 1  # S is float32, U float64: the kernel-list interpreter
    _both([Age, SampleExpr, pa.AdvectionRK4], spatial=np.float64)
    _both([SampleSpeed, pa.AdvectionRK2], mesh="spherical")


def test_sampling_outside_the_domain_marks_the_particle(gpu):
    """field.py:307-356: a sample outside the domain sets the error state like a built-in kernel's would; the user's recovery kernel
    behind it deletes the particle."""
    p, d = _both([pa.AdvectionRK4, SampleT, DeleteErrorParticle], margin=0.01, runtime=40 * 600.0, n=400)
    assert len(p) < 400


def test_boolean_algebra_and_numpy_functions(gpu):
    ctx = {"x0": 1.5e4, "y0": 1.0e4}
    _both([pa.AdvectionRK4, Age, Algebra], context=ctx)
    _both([pa.AdvectionRK4, Age, Algebra], context=ctx, spatial=np.float64)


def test_integer_variables(gpu):
    p, d = _both([pa.AdvectionRK4, Age, Counters])
    assert d["count"].dtype == np.int32 and d["flag"].dtype == np.int64 and set(np.unique(d["count"])) <= set(range(0, 14)) and (d["count"] == 7).any()
    _both([Counters, pa.AdvectionEE, Age], spatial=np.float64)


def test_kernels_that_change_dt_and_displacements(gpu):
    _both([pa.AdvectionRK4, Age, ChangeDt])
    _both([Kick, pa.AdvectionEE], context={"kick": 0.05})
    _both([Kick, pa.AdvectionEE], context={"kick": np.float32(0.05)}, spatial=np.float64)


def test_error_stop_with_an_accumulating_user_kernel(gpu):
    """Without a recovery kernel the run raises after the iteration of the first escape (kernel.py:236-245); the accumulated age of every
    particle is that of the iterations it made -- the compiled list restarts from the device checkpoint rather than repeating the launch
    on top of Variables it already updated in place."""
    p, d = _both([pa.AdvectionRK4, Age], margin=0.01, runtime=40 * 600.0, n=400, expect_error="FieldOutOfBoundError")
    assert p._last_stats["reran"] >= 1 and np.all(d["age"] <= d["t"].max() + 1)


def test_what_is_not_elementwise_runs_on_the_host(gpu):
    p, d = _run([pa.AdvectionRK4, NotElementwise], jit=True)
    assert p._kernel.user_program is None and "`if`" in p._kernel.jit_report and p._last_stats.get("hosted")
    assert np.all(d["acc"] > 0)

    def Closure(particles, fieldset):  # defined inside a function, with a free variable: still translatable
        particles.acc += scale * particles.dt

    scale = 0.5
    p, d = _both([Closure, pa.AdvectionEE])
    assert np.allclose(d["acc"][d["t"] == d["t"].max()].max(), 0.5 * 12 * 600.0)

    P = pa.Particle.add_variable(pa.Variable("count", dtype=np.int16, initial=0))

    def CountSteps(particles, fieldset):
        particles.count += 1

    fs = _fieldset()
    pset = pa.ParticleSet(fs, pclass=P, x=[2.0e4], y=[1.5e4])
    pset.execute([pa.AdvectionRK4, CountSteps], runtime=3000.0, dt=600.0)
    assert pset._kernel.user_program is None and "int16" in pset._kernel.jit_report and pset.count[0] == 5


def test_compiled_list_with_output_file(gpu, tmp_path):
    """The Variables a compiled kernel writes are device columns: the ParticleFile gets their values at every output time."""
    outs = []
    for jit in (True, False):
        path = tmp_path / f"out_{int(jit)}.parquet"
        pf = pa.ParticleFile(path, outputdt=1800.0)
        p, d = _run([pa.AdvectionRK4, Age, SampleT], jit=jit, output=pf, n=50)
        pf.close()
        outs.append(pa.read_particlefile(path).sort_values(["t", "particle_id"]).reset_index(drop=True))
    a, b = outs
    assert list(a.columns) == list(b.columns) and len(a) == len(b) and {"age", "temp"} <= set(a.columns)
    for c in a.columns:
        assert np.array_equal(a[c].values, b[c].values), c
    assert a["age"].max() > 0


def test_user_kernels_ride_in_the_dedicated_cgrid_kernel(gpu):
    """BASELINE config 3's shape (spherical curvilinear C-grid, AdvectionRK4_3D, populated `ei`) with an ageing kernel and the user's own
    recovery kernel around the advection kernel: the module carries advect_cgrid_kernel with the user kernels riding along
    (pk_exec_stats.program 101), bit-identical to the host path (the module is compiled with the exact quotients and full-range
    sines / cosines of the general program: parcels_amd/jit.py, PK_FAST_LEAN=0 PK_CG_NEAR=0 PK_CG_LEAN=0)."""
    from case_utils import build_fieldset
    from oracle import cases

    case = cases.curv_cgrid_case("jit_curv", mesh="spherical", kernels=["AdvectionRK4_3D"], seed=5, npart=600, nx=50, ny=40, vel=1.0, runtime=10 * 1800.0)
    out = []
    for jit in (True, False):
        os.environ["PARCELS_AMD_JIT"] = "1" if jit else "0"
        try:
            fs = build_fieldset(case)
            P = pa.get_default_particle(np.float64).add_variable(pa.Variable("age", dtype=np.float32, initial=0))
            pset = pa.ParticleSet(fs, pclass=P, x=np.asarray(case["x"]), y=np.asarray(case["y"]), z=np.asarray(case["z"]), t=np.zeros(len(case["x"])))
            pset.populate_indices()
            pset.execute([Age, pa.AdvectionRK4_3D, DeleteErrorParticle], runtime=float(case["runtime"]), dt=float(case["dt"]))
            out.append((pset, {k: np.array(v) for k, v in pset._data.items()}))
        finally:
            os.environ.pop("PARCELS_AMD_JIT", None)
    (pj, dj), (ph, dh) = out
    assert pj._kernel.user_program is not None and pj._kernel.user_program.flags == 1 and pj._last_stats["program"] == 101, pj._last_stats
    assert ph._kernel.user_program is None
    for k in dj:
        assert np.array_equal(dj[k], dh[k], equal_nan=True), k
    assert dj["age"].max() == 10 * 1800.0


def CosLat(particles, fieldset):
    particles.acc = np.cos(np.deg2rad(particles.y)) * np.exp(-particles.age / 5000)


def test_transcendental_functions_on_request(gpu, monkeypatch):
    """np.cos / np.exp in a kernel: by default the kernel keeps NumPy's values (host path); with PARCELS_AMD_JIT_LIBM=1 it is compiled and
    agrees with them to the ulp of the dtype the function ran in (float32 here for exp: `age` is a float32 Variable)."""
    monkeypatch.delenv("PARCELS_AMD_JIT_LIBM", raising=False)
    p0, d0 = _run([pa.AdvectionRK4, Age, CosLat], jit=True, mesh="spherical")
    assert p0._kernel.user_program is None and "PARCELS_AMD_JIT_LIBM" in p0._kernel.jit_report
    monkeypatch.setenv("PARCELS_AMD_JIT_LIBM", "1")
    p1, d1 = _run([pa.AdvectionRK4, Age, CosLat], jit=True, mesh="spherical")
    assert p1._kernel.user_program is not None
    for k in d0:
        if k == "acc":
            np.testing.assert_allclose(d1[k], d0[k], rtol=5e-7)
        else:
            assert np.array_equal(d1[k], d0[k], equal_nan=True), k


def SixVariables(particles, fieldset):
    particles.age += particles.dt
    particles.acc += particles.age
    particles.temp = particles.acc / 1000
    particles.speed = particles.dx - particles.dy
    particles.count += 1
    particles.flag = particles.count * 2 + particles.particle_id


def test_up_to_eight_user_variables_are_device_columns(gpu):
    p, d = _both([SixVariables, pa.AdvectionRK4])
    assert len(p._kernel.device_variables) == 6 and (d["count"] >= 12).any()


def MidpointAdvection(particles, fieldset):
    """The tutorials' hand-written RK2 (explanation_kernelloop.md, tutorial_nemo): the second sample at a computed point, particles attached."""
    u1, v1 = fieldset.UV[particles]
    x1 = particles.x + u1 * 0.5 * particles.dt
    y1 = particles.y + v1 * 0.5 * particles.dt
    u2, v2 = fieldset.UV[particles.t + 0.5 * particles.dt, particles.z, y1, x1, particles]
    particles.dx += u2 * particles.dt
    particles.dy += v2 * particles.dt


def SampleAhead(particles, fieldset):
    particles.temp = fieldset.T[particles.t, particles.z, particles.y, particles.x + fieldset.h, particles]


def GradientT(particles, fieldset):
    """Central difference of a tracer around the particle: samples WITHOUT the particles (value only; state and `ei` stay, field.py:173-176)."""
    east = fieldset.T[particles.t, particles.z, particles.y, particles.x + fieldset.h]
    west = fieldset.T[particles.t, particles.z, particles.y, particles.x - fieldset.h]
    particles.acc = (east - west) / (2 * fieldset.h)
    particles.temp = fieldset.T[particles.t, particles.z, particles.y, particles.x]


@pytest.mark.parametrize("mesh", ["flat", "spherical"])
@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_samples_at_computed_points(gpu, mesh, spatial):
    """fieldset.F[t, z, y, x, particles]: the hand-written mid-point scheme as the ONLY advection kernel (the kernel-list interpreter), and a
    look-ahead tracer sample riding in the dedicated A-grid kernel; on the spherical mesh the velocity conversion of the second sample is a
    float64 cosine (y1 is a float64 array) where the first one's is the particles' dtype."""
    h = 300.0 if mesh == "flat" else 0.25
    p, d = _both([MidpointAdvection, Age], mesh=mesh, spatial=spatial)
    assert np.abs(d["x"] - d["x"][0]).max() > 0
    p, d = _both([pa.AdvectionRK4, SampleAhead], mesh=mesh, spatial=spatial, context={"h": h})
    assert p._last_stats["program"] == 100 and p._kernel.user_program.flags & 1
    # leaving the domain through a sample at a computed point marks the particle like any other sample
    p, d = _both([pa.AdvectionRK4, SampleAhead, DeleteErrorParticle], mesh=mesh, spatial=spatial, context={"h": 40 * h}, margin=0.02, n=400)
    assert len(p) < 400


@pytest.mark.parametrize("mesh", ["flat", "spherical"])
def test_samples_without_the_particles(gpu, mesh):
    """fieldset.F[t, z, y, x]: neither the state nor `ei` changes -- a particle whose +-h neighbour lies outside the domain gets 0 from that
    side and goes on; the list runs in the kernel-list interpreter (the dedicated kernels keep state / ei where the restore cannot reach)."""
    h = 300.0 if mesh == "flat" else 0.25
    p, d = _both([pa.AdvectionRK4, GradientT], mesh=mesh, context={"h": h})
    assert p._kernel.user_program.sources[0].detached and not p._kernel.user_program.flags & 1 and np.abs(d["acc"]).max() > 0
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # (the host path warns about values masked to 0, field.py:359-370)
        p, d = _both([GradientT, pa.AdvectionEE], mesh=mesh, context={"h": 60 * h}, margin=0.02, n=200, runtime=3 * 600.0)
    assert len(p) == 200 and np.all(d["state"] < 50)


# ---- selections of the particles (the reference's ParticleSetView, particlesetview.py) ----------------------------------------------------
CYCLE_SPEED, DRIFT_DEPTH, MAX_DEPTH, MIN_DEPTH, DRIFT_TIME, CYCLE_TIME = 0.002, 2.0, 4.0, 0.5, 1800.0, 5400.0


def ArgoCycle(particles, fieldset):
    """tutorial_Argofloats.ipynb (its ArgoVerticalMovement, `count` as the phase): selections bound to locals, masks within them, a tracer
    sample at the positions of the floats in ONE phase."""
    ptcls0 = particles[particles.count == 0]
    ptcls1 = particles[particles.count == 1]
    ptcls2 = particles[particles.count == 2]
    ptcls3 = particles[particles.count == 3]
    ptcls4 = particles[particles.count == 4]
    ptcls0.dz += CYCLE_SPEED * ptcls0.dt
    next_phase = ptcls0.z + ptcls0.dz >= DRIFT_DEPTH
    ptcls0.count[next_phase] = 1
    ptcls0.dz[next_phase] = DRIFT_DEPTH - ptcls0.z[next_phase]
    ptcls1.age += ptcls1.dt
    next_phase = ptcls1.age >= DRIFT_TIME
    ptcls1.count[next_phase] = 2
    ptcls1.age[next_phase] = 0
    ptcls2.dz += CYCLE_SPEED * ptcls2.dt
    next_phase = ptcls2.z + ptcls2.dz >= MAX_DEPTH
    ptcls2.count[next_phase] = 3
    ptcls2.dz[next_phase] = MAX_DEPTH - ptcls2.z[next_phase]
    ptcls3.dz -= CYCLE_SPEED * ptcls3.dt
    ptcls3.temp = fieldset.T[ptcls3.t, ptcls3.z, ptcls3.y, ptcls3.x]
    next_phase = ptcls3.z + ptcls3.dz <= MIN_DEPTH
    ptcls3.count[next_phase] = 4
    ptcls3.dz[next_phase] = MIN_DEPTH - ptcls3.z[next_phase]
    next_phase = ptcls4.acc >= CYCLE_TIME
    ptcls4.count[next_phase] = 0
    ptcls4.acc[next_phase] = 0
    ptcls4.temp = np.nan
    particles.acc += particles.dt


def WarmWaterDrift(particles, fieldset):
    """tutorial_unstuck_Agrid.ipynb's SetDisplacement: a second sample only for the particles the first one selects."""
    particles.temp = fieldset.T[particles]
    warm = particles[particles.temp > 10.0]
    dU, dV = fieldset.UV[warm]
    warm.dx += dU * warm.dt * 0.5
    warm.dy += dV * warm.dt * 0.5
    particles[particles.temp <= 10.0].age += 1.0


def BounceBack(particles, fieldset):
    """tests/test_particleset_execute.py's MoveLeft / tutorial_statuscodes.md: recovery through an index selection."""
    inds = np.where(particles.state == StatusCode.ErrorOutOfBounds)
    particles[inds].dx = -particles[inds].dx
    particles[inds].dy = -particles[inds].dy
    particles[inds].count += 1
    particles[inds].state = StatusCode.Evaluate


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_argo_cycle_over_selections(gpu, spatial):
    p, d = _both([ArgoCycle, pa.AdvectionRK4], spatial=spatial, runtime=40 * 600.0)
    prog = p._kernel.user_program
    assert prog.sources[0].counter is not None and prog.sources[0].detached
    assert (d["acc"] > 0).all() and (d["z"] != 0).any()  # the floats moved through their phases


@pytest.mark.parametrize("mesh", ["flat", "spherical"])
def test_a_sample_for_the_particles_an_earlier_sample_selects(gpu, mesh):
    p, d = _both([pa.AdvectionRK4, WarmWaterDrift], mesh=mesh)
    assert p._kernel.user_program.sources[0].counter is not None and p._last_stats["program"] == 100  # rides in the dedicated A-grid kernel
    assert 0 < (d["age"] > 0).sum() < len(d["age"])  # some particles were warm, some were not
    _both([WarmWaterDrift, pa.AdvectionEE, Age], mesh=mesh, spatial=np.float64)


def test_recovery_through_an_index_selection(gpu):
    p, d = _both([pa.AdvectionRK4, BounceBack], margin=0.01, runtime=60 * 600.0, n=400)
    assert len(p) == 400 and d["count"].max() >= 1
