"""CPU: the host-side restatement of Kernel.execute (parcels_amd/hostkernels.py::execute_hosted, kernel.py:174-247 of the reference)
against the reference's REAL loop.  A kernel list made of Python functions alone needs no device, so the same functions run (a) through the
reference's own ParticleSet / Kernel.execute (loaded unmodified under oracle/ref_shim.py) and (b) through execute_hosted on parcels_amd's
ParticleSet -- dt clipping towards the end time, the evaluate mask, position update, dt reset, EndofLoop, deletion inside the loop, the
StopExecution / StopAllExecution states, error codes raised in ErrorsToThrow order, backward runs, staggered release times -- and leave
the same columns, bit for bit."""
import numpy as np
import pytest

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")


def _fieldsets():
    """A tiny zero-velocity FieldSet on both sides (the kernels below never sample it)."""
    import parcels_amd as pa
    from case_utils import build_fieldset
    from oracle import cases
    from oracle.make_golden import build_ref_fieldset

    case = cases.rect_agrid_case("loop", mesh="flat", kernels=["AdvectionRK4"], seed=1, npart=4, nx=6, ny=5, nz=2, nt=2)
    ref_fs, _ = build_ref_fieldset(case)
    return ref_fs, build_fieldset(case), pa


def Drift(particles, fieldset):
    particles.dx += 0.25 * particles.dt
    particles.dy -= 0.125 * particles.dt
    particles.age += particles.dt


def DeleteFar(particles, fieldset):
    particles.state = np.where(np.abs(particles.x) > fieldset.far, 30, particles.state)  # StatusCode.Delete


def HalveDt(particles, fieldset):
    # (only for a while: halving dt all the way to the end time never arrives -- the reference's loop and its restatement both spin)
    particles.dt = np.where((np.abs(particles.age) > 2500) & (np.abs(particles.age) < 4000), particles.dt / 2, particles.dt)


def StopSome(particles, fieldset):
    particles.state = np.where((np.mod(particles.particle_id, 5) == 0) & (np.abs(particles.age) >= 1800), 40, particles.state)  # StatusCode.StopExecution


def StopAll(particles, fieldset):
    particles.state = np.where(np.abs(particles.age) >= 3600, 41, particles.state)  # StatusCode.StopAllExecution


def RaiseOutOfBounds(particles, fieldset):
    particles.state = np.where((particles.particle_id == 3) & (np.abs(particles.age) >= 1200), 60, particles.state)  # ErrorOutOfBounds


def RaiseTwo(particles, fieldset):
    particles.state = np.where((particles.particle_id == 2) & (np.abs(particles.age) >= 600), 70, particles.state)  # ErrorOutsideTimeInterval ...
    particles.state = np.where((particles.particle_id == 4) & (np.abs(particles.age) >= 600), 51, particles.state)  # ... and ErrorInterpolation


LISTS = {
    "drift": [Drift],
    "delete": [Drift, DeleteFar],
    "dt": [Drift, HalveDt],
    "stop_some": [Drift, StopSome],
    "stop_all": [Drift, StopAll],
    "error": [Drift, RaiseOutOfBounds],
    "two_errors": [RaiseTwo, Drift],
}


@pytest.mark.parametrize("which", sorted(LISTS))
@pytest.mark.parametrize("direction", [1, -1])
@pytest.mark.parametrize("spatial", [np.float32, np.float64])
def test_host_loop_equals_the_references_loop(which, direction, spatial):
    ref_fs, my_fs, pa = _fieldsets()
    for fs in (ref_fs, my_fs):
        fs.add_context("far", 1500.0)
    m = ref_shim.load_reference()
    n = 12
    rng = np.random.default_rng(7)
    x, y = rng.uniform(100, 900, n), rng.uniform(100, 900, n)
    t0 = np.where(np.arange(n) % 3 == 0, 600.0, 0.0) * (1 if direction > 0 else 0) + (7200.0 if direction < 0 else 0.0)
    dt, endtime = 600.0 * direction, (7200.0 if direction > 0 else 0.0)
    funcs = LISTS[which]

    # (a) the reference: its own particle class, ParticleSet and Kernel
    RP = m["particle"]
    rclass = RP.get_default_particle(spatial).add_variable([RP.Variable("age", dtype=np.float32, initial=0)])
    rset = m["particleset"].ParticleSet(ref_fs, pclass=rclass, x=x, y=y, z=np.zeros(n), t=(t0 * 1e9).round().astype("int64").astype("timedelta64[ns]"))
    rset._data["dt"][:] = dt
    rk = m["kernel"].Kernel(list(funcs), rset)
    rerr = None
    try:
        rres = rk.execute(rset, endtime, dt)
    except Exception as e:
        rerr, rres = type(e).__name__, None

    # (b) parcels_amd: its ParticleSet and the host restatement of the loop
    from parcels_amd.hostkernels import execute_hosted
    from parcels_amd.kernel import Kernel

    pclass = pa.get_default_particle(spatial).add_variable([pa.Variable("age", dtype=np.float32, initial=0)])
    pset = pa.ParticleSet(my_fs, pclass=pclass, x=x, y=y, z=np.zeros(n), t=t0)
    pset._data["dt"][:] = dt
    k = Kernel(list(funcs), pset)
    merr = None
    try:
        execute_hosted(k, pset, endtime, dt)
    except Exception as e:
        merr = type(e).__name__
    assert merr == rerr, (merr, rerr)
    if which == "stop_all":
        assert int(rres) == 41
    a, b = rset._data, pset._data
    assert set(a) == set(b)
    for key in a:
        assert a[key].dtype == b[key].dtype and np.array_equal(a[key], b[key], equal_nan=True), (which, key, a[key], b[key])
    if which == "delete":
        assert len(b["x"]) < n or direction < 0
