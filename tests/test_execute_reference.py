"""The reference's tests/test_particleset_execute.py and tests/test_kernel.py, restated against parcels_amd for every test whose
kernel is a built-in or one of the toy kernels of tests/common_kernels.py (DoNothing, MoveEast, MoveNorth -- native tokens here).
Tests built on arbitrary Python kernel bodies have no counterpart (the binding falls through to the reference's
NumPy loop for those); where the reference uses such a body only as a step counter, the closest native kernel stands in and the
expected numbers are scaled accordingly (said at each test)."""

from contextlib import nullcontext as does_not_raise
from datetime import datetime, timedelta

import numpy as np
import pytest

import parcels_amd as pa
from parcels_amd.field import to_seconds
from test_particlefile_reference import make_fieldset


@pytest.fixture
def fieldset():
    return make_fieldset()


@pytest.fixture
def fieldset_no_time_interval():
    return make_fieldset(time=None)


@pytest.fixture
def zonal_flow_fieldset():
    return make_fieldset(uniform=(1.0, 0.0))


# ---- tests/test_particleset_execute.py ------------------------------------------------------------------------------------------
def test_pset_execute_invalid_arguments(fieldset, fieldset_no_time_interval):  # :98-149 (argument validation: before any launch)
    def pset(fs=fieldset):
        return pa.ParticleSet(fs, x=[0.2], y=[1.0])

    with pytest.raises(ValueError, match="dt must be a non-zero datetime.timedelta or np.timedelta64 object, got .*"):
        pset().execute(pa.AdvectionRK4, dt=np.timedelta64(0, "s"))
    with pytest.raises(ValueError, match="runtime and endtime are mutually exclusive - provide one or the other. Got .*"):
        pset().execute(pa.AdvectionRK4, runtime=np.timedelta64(1, "s"), endtime=np.datetime64("2100-01-01"), dt=np.timedelta64(1, "s"))
    msg = "Calculated/provided end time of .* is not in fieldset time interval .* Either reduce your runtime, modify your provided endtime, or change your release timing.*"
    with pytest.raises(ValueError, match=msg):
        pset().execute(pa.AdvectionRK4, endtime=np.datetime64("1990-01-01"), dt=np.timedelta64(1, "s"))
    with pytest.raises(ValueError, match=msg):
        pset().execute(pa.AdvectionRK4, endtime=np.datetime64("2100-01-01"), dt=np.timedelta64(-1, "s"))
    with pytest.raises(ValueError, match="The endtime must be of the same type as the fieldset.time_interval start time. Got .*"):
        pset().execute(pa.AdvectionRK4, endtime=12345, dt=np.timedelta64(1, "s"))
    with pytest.raises(ValueError, match="The runtime must be provided when the time_interval is not defined for a fieldset."):
        pset(fieldset_no_time_interval).execute(pa.AdvectionRK4, dt=np.timedelta64(1, "s"))


@pytest.mark.gpu
@pytest.mark.parametrize("runtime, expectation", [  # :151-163
    (np.timedelta64(5, "s"), does_not_raise()),
    (timedelta(seconds=2), does_not_raise()),
    (5.0, does_not_raise()),
    (np.datetime64("2001-01-02T00:00:00"), pytest.raises(ValueError)),
    (datetime(2000, 1, 2, 0, 0, 0), pytest.raises(ValueError)),
])
def test_particleset_runtime_type(gpu, fieldset, runtime, expectation):
    pset = pa.ParticleSet(fieldset, x=[0.2], y=[1.0])
    with expectation:
        pset.execute(runtime=runtime, dt=np.timedelta64(10, "s"), kernels=pa.DoNothing)


@pytest.mark.gpu
@pytest.mark.parametrize("endtime, expectation", [  # :166-179
    (np.datetime64("2000-01-02T00:00:00"), does_not_raise()),
    (5.0, pytest.raises(ValueError)),
    (np.timedelta64(5, "s"), pytest.raises(ValueError)),
    (timedelta(seconds=2), pytest.raises(ValueError)),
    (datetime(2000, 1, 2, 0, 0, 0), pytest.raises(ValueError)),
])
def test_particleset_endtime_type(gpu, fieldset, endtime, expectation):
    pset = pa.ParticleSet(fieldset, x=[0.2], y=[1.0])
    with expectation:
        pset.execute(endtime=endtime, dt=np.timedelta64(10, "m"), kernels=pa.DoNothing)


def test_sampleUonly(fieldset):  # :182-192: sampling a velocity component on its own warns (the kernel is the SampleField token)
    pclass = pa.get_default_particle(np.float32).add_variable(pa.Variable("u", dtype=np.float32, initial=0.0))
    pset = pa.ParticleSet(fieldset, pclass=pclass, x=[0.2], y=[1.0])
    with pytest.warns(RuntimeWarning, match="Sampling of velocities should normally be done using fieldset.UV or fieldset.UVW object; tread carefully"):
        pa.Kernel([pa.SampleField("U", into="u")], pset)


@pytest.mark.gpu
def test_particleset_run_to_endtime(gpu, fieldset):  # :195-204
    starttime, endtime = fieldset.time_interval.left, fieldset.time_interval.right
    pset = pa.ParticleSet(fieldset, x=[0.2], y=[1.0], t=[starttime])
    pset.execute(pa.DoNothing, endtime=endtime, dt=np.timedelta64(1, "D"))
    assert np.timedelta64(int(pset.t[0]), "s") + fieldset.time_interval.left == endtime


@pytest.mark.gpu
def test_particleset_interpolate_outside_domainedge(gpu, zonal_flow_fieldset):  # :246-257
    fs = zonal_flow_fieldset
    pset = pa.ParticleSet(fs, x=fs.U.grid.lon[-1], y=fs.U.grid.lat[-1] + 1e-3)
    with pytest.raises(pa.FieldOutOfBoundError):
        pset.execute(pa.AdvectionEE, runtime=np.timedelta64(2, "D"), dt=np.timedelta64(1, "D"))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.timedelta64(1, "s"), np.timedelta64(1, "ms"), np.timedelta64(10, "ms"), np.timedelta64(1, "ns")])
def test_pset_execute_subsecond_dt(gpu, fieldset, dt):  # :259-269 (AddDt accumulates particles.dt: here the clock itself is read)
    pset = pa.ParticleSet(fieldset, x=0, y=0)
    pset.execute(pa.DoNothing, runtime=dt * 10, dt=dt)
    np.testing.assert_allclose(pset.t[0], 10.0 * to_seconds(dt), atol=1e-5 * to_seconds(dt))
    assert pset.dt[0] == to_seconds(dt)


@pytest.mark.gpu
@pytest.mark.parametrize("with_delete", [True, False])
def test_pset_multi_execute(gpu, fieldset, with_delete, npart=10, n=5):  # :298-311 (AddLat == MoveNorth)
    pset = pa.ParticleSet(fieldset, pclass=pa.get_default_particle(np.float64), x=np.linspace(0, 1, npart), y=np.zeros(npart))
    for _ in range(n):
        pset.execute(pa.MoveNorth, runtime=np.timedelta64(1, "s"), dt=np.timedelta64(1, "s"))
        if with_delete:
            pset.remove_indices(len(pset) - 1)
    assert len(pset) == (npart - n if with_delete else npart)
    np.testing.assert_allclose(pset.y, n * 0.1, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("starttime, endtime, dt", [(0, 10, 1), (0, 10, 3), (2, 16, 3), (20, 10, -1), (20, 0, -2), (5, 15, 1)])
def test_execution_endtime(gpu, fieldset, starttime, endtime, dt):  # :314-325
    starttime = fieldset.time_interval.left + np.timedelta64(starttime, "s")
    endtime = fieldset.time_interval.left + np.timedelta64(endtime, "s")
    pset = pa.ParticleSet(fieldset, t=starttime, x=0, y=0)
    pset.execute(pa.DoNothing, endtime=endtime, dt=np.timedelta64(dt, "s"))
    assert pset.t == to_seconds(endtime - fieldset.time_interval.left)


@pytest.mark.gpu
def test_dont_run_particles_outside_starttime(gpu, fieldset):  # :328-356 (AddLon adds 1 per step; MoveEast adds 0.1)
    fs = fieldset
    start_times = [fs.time_interval.left + np.timedelta64(t, "s") for t in [0, 2, 10]]
    endtime = fs.time_interval.left + np.timedelta64(8, "s")
    pset = pa.ParticleSet(fs, pclass=pa.get_default_particle(np.float64), x=np.zeros(3), y=np.zeros(3), t=start_times)
    pset.execute(pa.MoveEast, dt=np.timedelta64(1, "s"), endtime=endtime)
    np.testing.assert_allclose(pset.x, [0.8, 0.6, 0], atol=1e-12)
    assert pset.t[0] == pset.t[1] == to_seconds(endtime - fs.time_interval.left)
    assert pset.t[2] == to_seconds(start_times[2] - fs.time_interval.left)  # this particle has not been executed
    start_times = [fs.time_interval.right - np.timedelta64(t, "s") for t in [0, 2, 10]]
    endtime = fs.time_interval.right - np.timedelta64(8, "s")
    pset = pa.ParticleSet(fs, pclass=pa.get_default_particle(np.float64), x=np.zeros(3), y=np.zeros(3), t=start_times)
    pset.execute(pa.MoveEast, dt=-np.timedelta64(1, "s"), endtime=endtime)
    np.testing.assert_allclose(pset.x, [0.8, 0.6, 0], atol=1e-12)
    assert pset.t[0] == pset.t[1] == to_seconds(endtime - fs.time_interval.left)
    assert pset.t[2] == to_seconds(start_times[2] - fs.time_interval.left)


@pytest.mark.gpu
def test_some_particles_throw_outofbounds(gpu, zonal_flow_fieldset):  # :359-365 (flat mesh, U = 1: the eastern particles leave)
    lon = np.linspace(0, 3.9, 100)
    pset = pa.ParticleSet(zonal_flow_fieldset, x=lon, y=np.zeros_like(lon))
    with pytest.raises(pa.FieldOutOfBoundError):
        pset.execute(pa.AdvectionEE, runtime=np.timedelta64(100, "s"), dt=np.timedelta64(1, "s"))


@pytest.mark.gpu
def test_some_particles_throw_outoftime(gpu, fieldset):  # :381-389
    """The reference's kernel samples UV 400 days ahead of the particle clock.  With built-in kernels a field is only ever sampled at
    the particle's own time (+ RK stage offsets), so the out-of-time sample is produced by a release before the first time level."""
    time = [fieldset.time_interval.left + np.timedelta64(t, "h") for t in [0, -1]]
    with pytest.warns(pa.ParticleSetWarning):
        pset = pa.ParticleSet(fieldset, x=np.zeros(2), y=np.zeros(2), t=time)
    with pytest.raises(pa.OutsideTimeInterval):
        pset.execute(pa.AdvectionEE, runtime=np.timedelta64(2, "h"), dt=np.timedelta64(10, "m"))


@pytest.mark.gpu
@pytest.mark.parametrize("starttime_flt, runtime_flt, dt", [(0, 10, 1), (0, 10, 3), (2, 16, 3), (20, 10, -1), (20, 0, -2), (5, 15, 1)])
@pytest.mark.parametrize("npart", [1, 10])
def test_execution_runtime(gpu, fieldset, starttime_flt, runtime_flt, dt, npart):  # :445-457
    starttime = fieldset.time_interval.left + np.timedelta64(starttime_flt, "s")
    pset = pa.ParticleSet(fieldset, t=starttime, x=np.zeros(npart), y=np.zeros(npart))
    pset.execute(pa.DoNothing, runtime=np.timedelta64(runtime_flt, "s"), dt=np.timedelta64(dt, "s"))
    assert np.all(np.abs(pset.t - starttime_flt - runtime_flt * np.sign(dt)) < 1e-3)


@pytest.mark.gpu
def test_changing_dt_in_kernel(gpu, fieldset):  # :460-468 (KernelCounter adds 1 per execution; MoveEast adds 0.1): 3 executions
    pset = pa.ParticleSet(fieldset, pclass=pa.get_default_particle(np.float64), x=np.zeros(1), y=np.zeros(1))
    pset.execute(pa.MoveEast, dt=np.timedelta64(2, "s"), runtime=np.timedelta64(5, "s"))
    np.testing.assert_allclose(pset.x, 0.3, atol=1e-12)
    assert pset.dt == 2  # the shortened last step (1 s) does not stick
    assert pset.t == 5


# ---- tests/test_kernel.py -------------------------------------------------------------------------------------------------------
def test_kernel_init(fieldset):  # :54-56
    pa.Kernel(kernels=[pa.AdvectionRK4], pset=pa.ParticleSet(fieldset, x=[0.5], y=[0.5]))


def test_kernel_merging(fieldset):  # :59-69
    pset = pa.ParticleSet(fieldset, x=[0.5], y=[0.5])
    merged = pa.Kernel(kernels=[pa.AdvectionRK4, pa.MoveEast, pa.MoveNorth], pset=pset)
    assert merged.funcname == "AdvectionRK4MoveEastMoveNorth"
    assert merged._kernels == [pa.AdvectionRK4, pa.MoveEast, pa.MoveNorth]
    merged = pa.Kernel(kernels=[pa.MoveEast, pa.MoveNorth, pa.AdvectionRK4], pset=pset)
    assert merged.funcname == "MoveEastMoveNorthAdvectionRK4"
    assert merged._kernels == [pa.MoveEast, pa.MoveNorth, pa.AdvectionRK4]


def test_kernel_from_list_error_checking(fieldset):  # :88-104
    pset = pa.ParticleSet(fieldset, x=[0.5], y=[0.5])
    with pytest.raises(ValueError, match="List of `kernels` should have at least one function."):
        pa.Kernel(kernels=[], pset=pset)
    with pytest.raises(TypeError, match=r"Argument `kernels` should be a function or list of functions.*"):
        pa.Kernel(kernels=[pa.AdvectionRK4, "something else"], pset=pset)
    with pytest.raises(TypeError, match=r".* should be a function or list of functions.*"):
        pa.Kernel(kernels=[pa.Kernel(kernels=[pa.AdvectionRK4], pset=pset), pa.MoveEast, pa.MoveNorth], pset=pset)


def test_RK45Kernel_error_no_next_dt(fieldset):  # :107-112
    pset = pa.ParticleSet(fieldset, x=[0.5], y=[0.5])
    with pytest.raises(ValueError, match='ParticleClass requires a "next_dt" for AdvectionRK45 Kernel.'):
        pa.Kernel(kernels=[pa.AdvectionRK45], pset=pset)


@pytest.mark.gpu
def test_rk45_kernel_warnings(gpu, fieldset):  # :115-124
    pclass = pa.Particle.add_variable(pa.Variable("next_dt", dtype=np.float32, initial=1))
    pset = pa.ParticleSet(fieldset=fieldset, pclass=pclass, x=[0], y=[0], next_dt=1)
    with pytest.warns(pa.KernelWarning):
        pset.execute(pa.AdvectionRK45, runtime=1, dt=1)


def test_kernel_signature(fieldset):  # :127-165 (the cases that do not need a Python body to run)
    pset = pa.ParticleSet(fieldset, x=[0.5], y=[0.5])

    def kernel_switched_args(fieldset, particle):
        pass

    def kernel_with_forced_kwarg(particles, *, fieldset=0):
        pass

    for bad in (kernel_switched_args, kernel_with_forced_kwarg):
        with pytest.raises(ValueError):
            pa.Kernel(kernels=[bad], pset=pset)


@pytest.mark.gpu
def test_execution_order(gpu):  # :167-202, kernel_type "update_dlon": the order of kernels writing dx / dy does not matter
    lons, lats = [], []
    for direction in (1, -1):
        pset = pa.ParticleSet(make_fieldset(), pclass=pa.get_default_particle(np.float64), x=0, y=0)
        pset.execute([pa.MoveEast, pa.AdvectionRK4, pa.MoveNorth][::direction], runtime=1, dt=1)
        lons.append(pset.x[0])
        lats.append(pset.y[0])
    assert lons[0] == lons[1] and lats[0] == lats[1]
    assert np.isclose(lons[0], 0.1, atol=1e-4) and np.isclose(lats[0], 0.1, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [1e-2, 1e-5, 1e-6, 1e-9])
def test_small_dt(gpu, fieldset, dt):  # :221-226
    pset = pa.ParticleSet(fieldset, x=[0], y=[0])
    pset.execute(pa.DoNothing, dt=dt, runtime=dt * 100)
    assert np.allclose(pset.t, dt * 100)
