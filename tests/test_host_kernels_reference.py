"""The reference's tests with HAND-WRITTEN Python kernels (tests/test_particleset_execute.py: the `def kernel(particles, fieldset)`
plug-in point, kernel.py:67-70 and its loop :190-245), restated one for one against parcels_amd: a kernel list that contains a Python
function runs the reference's loop on the host columns (parcels_amd/hostkernels.py) with the built-in kernels' bodies and all field
sampling still on the GPU.  Line numbers refer to the reference's test file."""

import numpy as np
import pytest

import parcels_amd as pa
from parcels_amd import StatusCode
from test_particlefile_reference import make_fieldset

pytestmark = pytest.mark.gpu


@pytest.fixture
def fieldset():
    return make_fieldset()


def test_sampling_in_a_python_kernel(gpu, fieldset):  # :182-205, :230-243
    def SampleU(particles, fieldset):
        _ = fieldset.U[particles]

    pset = pa.ParticleSet(fieldset, x=[0.2], y=[1.0])
    with pytest.warns(RuntimeWarning, match="Sampling of velocities should normally be done using fieldset.UV or fieldset.UVW object; tread carefully"):
        pset.execute(SampleU, runtime=np.timedelta64(1, "D"), dt=np.timedelta64(1, "D"))

    def SampleUV(particles, fieldset):
        particles.var, _ = fieldset.UV[particles]

    fs = make_fieldset(uniform=(1.0, 0.0))
    MyParticle = pa.Particle.add_variable(pa.Variable("var", dtype=np.float64, initial=0))
    pset = pa.ParticleSet(fs, pclass=MyParticle, x=[float(fs.U.grid.lon[3])], y=[float(fs.U.grid.lat[3])])
    pset.execute(SampleUV, runtime=np.timedelta64(1, "s"), dt=np.timedelta64(1, "s"))
    assert pset[0].var == 1.0 and pset[0].t == 1.0


@pytest.mark.parametrize("dt", [np.timedelta64(1, "s"), np.timedelta64(1, "ms"), np.timedelta64(10, "ms"), np.timedelta64(1, "ns")])
def test_pset_execute_subsecond_dt(gpu, fieldset, dt):  # :259-269
    def AddDt(particles, fieldset):
        particles.added_dt += particles.dt

    pclass = pa.Particle.add_variable(pa.Variable("added_dt", dtype=np.float32, initial=0))
    pset = pa.ParticleSet(fieldset, pclass=pclass, x=0, y=0)
    pset.execute(AddDt, runtime=dt * 10, dt=dt)
    np.testing.assert_allclose(pset[0].added_dt, 10.0 * float(dt / np.timedelta64(1, "s")), atol=1e-5)


def test_pset_remove_particle_in_kernel(gpu, fieldset):  # :272-284
    npart = 100
    pset = pa.ParticleSet(fieldset, x=np.linspace(0, 1, npart), y=np.linspace(1, 0, npart))

    def DeleteKernel(particles, fieldset):
        particles.state = np.where((particles.x >= 0.4) & (particles.x <= 0.6), StatusCode.Delete, particles.state)

    pset.execute(DeleteKernel, runtime=np.timedelta64(1, "s"), dt=np.timedelta64(1, "s"))
    indices = [i for i in range(npart) if not (40 <= i < 60)]
    assert [p.particle_id for p in pset] == indices
    assert pset[70].particle_id == 90 and pset[-1].particle_id == npart - 1 and pset.size == 80


@pytest.mark.parametrize("npart", [1, 100])
def test_pset_stop_simulation(gpu, fieldset, npart):  # :287-295
    pset = pa.ParticleSet(fieldset, x=np.zeros(npart), y=np.zeros(npart), pclass=pa.Particle)

    def Delete(particles, fieldset):
        particles[particles.t >= 4].state = StatusCode.StopExecution

    pset.execute(Delete, dt=np.timedelta64(1, "s"), runtime=np.timedelta64(21, "s"))
    assert pset[0].t == 4


@pytest.mark.parametrize("with_delete", [True, False])
def test_pset_multi_execute(gpu, fieldset, with_delete, npart=10, n=5):  # :298-313
    pset = pa.ParticleSet(fieldset, x=np.linspace(0, 1, npart), y=np.zeros(npart))

    def AddLat(particles, fieldset):
        particles.dy += 0.1

    for _ in range(n):
        pset.execute(AddLat, runtime=np.timedelta64(1, "s"), dt=np.timedelta64(1, "s"))
        if with_delete:
            pset.remove_indices(len(pset) - 1)
    if with_delete:
        assert np.allclose(pset.y, n * 0.1, atol=1e-6)
    else:
        assert np.allclose([p.y - n * 0.1 for p in pset], np.zeros(npart), atol=1e-6)


def test_dont_run_particles_outside_starttime(gpu, fieldset):  # :329-356
    left, right = fieldset.time_interval.left, fieldset.time_interval.right

    def AddLon(particles, fieldset):
        particles.x += 1

    start_times = [left + np.timedelta64(t, "s") for t in [0, 2, 10]]
    endtime = left + np.timedelta64(8, "s")
    pset = pa.ParticleSet(fieldset, x=np.zeros(3), y=np.zeros(3), t=start_times)
    pset.execute(AddLon, dt=np.timedelta64(1, "s"), endtime=endtime)
    np.testing.assert_array_equal(pset.x, [8, 6, 0])
    assert pset.t[0] == 8.0 and pset.t[2] == 10.0  # the third particle has not been executed
    start_times = [right - np.timedelta64(t, "s") for t in [0, 2, 10]]
    endtime = right - np.timedelta64(8, "s")
    pset = pa.ParticleSet(fieldset, x=np.zeros(3), y=np.zeros(3), t=start_times)
    pset.execute(AddLon, dt=-np.timedelta64(1, "s"), endtime=endtime)
    np.testing.assert_array_equal(pset.x, [8, 6, 0])


def test_delete_on_all_errors(gpu, fieldset):  # :368-378
    def MoveRight(particles, fieldset):
        particles.dx += 1
        fieldset.UV[particles.t, particles.z, particles.y, particles.x, particles]

    def DeleteAllErrorParticles(particles, fieldset):
        particles[particles.state > 20].state = StatusCode.Delete

    pset = pa.ParticleSet(fieldset, x=[1e5, 2], y=[0, 0])
    pset.execute([MoveRight, DeleteAllErrorParticles], runtime=np.timedelta64(10, "s"), dt=np.timedelta64(1, "s"))
    assert len(pset) == 0


def test_some_particles_throw_outoftime(gpu, fieldset):  # :381-389
    left = fieldset.time_interval.left
    time = [left + np.timedelta64(t, "D") for t in [0, 350]]
    pset = pa.ParticleSet(fieldset, x=np.zeros(2), y=np.zeros(2), t=time)

    def FieldAccessOutsideTime(particles, fieldset):
        fieldset.UV[particles.t + 400 * 86400, particles.z, particles.y, particles.x, particles]

    with pytest.raises(pa.OutsideTimeInterval):
        pset.execute(FieldAccessOutsideTime, runtime=np.timedelta64(1, "D"), dt=np.timedelta64(10, "D"))


def test_execution_check_stopallexecution(gpu, fieldset):  # :413-421
    def addoneLon(particles, fieldset):
        particles.dx += 1
        particles[particles.x + particles.dx >= 10].state = StatusCode.StopAllExecution

    pset = pa.ParticleSet(fieldset, x=[0, 0], y=[0, 0])
    pset.execute(addoneLon, runtime=np.timedelta64(20, "s"), dt=np.timedelta64(1, "s"))
    np.testing.assert_allclose(pset.x, 9)
    np.testing.assert_allclose(pset.t, 9)


def test_execution_recover_out_of_bounds(gpu, fieldset):  # :424-444 (the domain of THIS fieldset is lon -2..4: expected values by the same rule)
    npart = 2

    def MoveRight(particles, fieldset):
        fieldset.UV[particles.t, particles.z, particles.y, particles.x + 0.1, particles]
        particles.dx += 0.1

    def MoveLeft(particles, fieldset):
        inds = np.where(particles.state == StatusCode.ErrorOutOfBounds)
        particles[inds].dx -= 1.0
        particles[inds].state = StatusCode.Success

    lon = np.array([-1.93, 3.93])
    lat = np.linspace(1, 0, npart)
    pset = pa.ParticleSet(fieldset, x=lon, y=lat)
    pset.execute([MoveRight, MoveLeft], runtime=np.timedelta64(60, "s"), dt=np.timedelta64(1, "s"))
    assert len(pset) == npart
    expect = lon.copy()
    for _ in range(60):  # a step of +0.1, or of 0.1 - 1.0 when the sample point x + 0.1 lies beyond the last node (4.0)
        expect += np.where(expect + 0.1 > 4.0, 0.1 - 1.0, 0.1)
    assert np.allclose(expect, [3.07, 3.93])  # the first particle bounced once, the second six times
    np.testing.assert_allclose(pset.x, expect, rtol=1e-5)
    np.testing.assert_allclose(pset.y, lat, rtol=1e-5)


def test_changing_dt_in_kernel(gpu, fieldset):  # :460-469
    def KernelCounter(particles, fieldset):
        particles.x += 1

    pset = pa.ParticleSet(fieldset, x=np.zeros(1), y=np.zeros(1))
    pset.execute(KernelCounter, dt=np.timedelta64(2, "s"), runtime=np.timedelta64(5, "s"))
    assert pset.x == 3 and pset.dt == 2 and pset.t == 5


@pytest.mark.parametrize("npart", [1, 100])
def test_execution_fail_python_exception(gpu, fieldset, npart):  # :472-484
    pset = pa.ParticleSet(fieldset, x=np.linspace(0, 1, npart), y=np.linspace(1, 0, npart))

    def PythonFail(particles, fieldset):
        inds = np.argwhere(particles.t >= 10)
        if inds.size > 0:
            raise RuntimeError("Enough is enough!")

    with pytest.raises(RuntimeError):
        pset.execute(PythonFail, runtime=np.timedelta64(20, "s"), dt=np.timedelta64(2, "s"))
    assert len(pset) == npart and all(pset.t == 10)


@pytest.mark.parametrize("kernel_names, expected", [("Lat1", [0, 1]), ("Lat2", [2, 0]), ("Lat1and2", [2, 1]), ("Lat1then2", [2, 1])])
def test_execution_update_particle_in_kernel_function(gpu, fieldset, kernel_names, expected):  # :487-531
    pset = pa.ParticleSet(fieldset, x=np.linspace(0, 1, 2), y=np.zeros(2))

    def Lat1(particles, fieldset):
        def SetLat1(p):
            p.y = 1

        SetLat1(particles[(particles.y == 0) & (particles.x > 0.5)])

    def Lat2(particles, fieldset):
        def SetLat2(p):
            p.y = 2

        SetLat2(particles[(particles.y == 0) & (particles.x < 0.5)])

    def Lat1and2(particles, fieldset):
        Lat1(particles, fieldset)
        Lat2(particles, fieldset)

    kernels = {"Lat1": [Lat1], "Lat2": [Lat2], "Lat1and2": [Lat1and2], "Lat1then2": [Lat1, Lat2]}[kernel_names]
    pset.execute(kernels, runtime=np.timedelta64(2, "s"), dt=np.timedelta64(1, "s"))
    np.testing.assert_allclose(pset.y, expected, rtol=1e-5)


def test_python_kernel_between_device_kernels_equals_the_fused_run(gpu):
    """[AdvectionRK4, Age, DeleteParticle] -- a Python kernel between two device kernels, float32 particles -- gives the positions
    of the fused [AdvectionRK4, DeleteParticle] launch bit for bit (the device bodies are the same code, the host does the loop), and
    the ages the loop implies; particles that leave the domain are deleted by the built-in kernel AFTER the Python one saw them."""
    from case_utils import build_fieldset, build_pset
    from oracle import cases

    case = cases.rect_agrid_case("hosted", mesh="flat", kernels=["AdvectionRK4", "DeleteParticle"], seed=5, npart=400, vel=4.0, margin=0.02, dt=1800.0,
                                 runtime=12 * 1800.0, spatial_dtype="float32")
    fused = build_pset(case, build_fieldset(case))
    fused.execute([pa.AdvectionRK4, pa.DeleteParticle], dt=1800.0, runtime=case["runtime"])
    seen_errors = []

    def Age(particles, fieldset):
        particles.age += particles.dt
        seen_errors.append(int(np.sum(particles.state >= StatusCode.Error)))

    fs = build_fieldset(case)
    pclass = pa.get_default_particle(np.float32).add_variable(pa.Variable("age", dtype=np.float64, initial=0))
    hosted = pa.ParticleSet(fs, pclass=pclass, x=case["x"], y=case["y"], z=case["z"], t=np.zeros(len(case["x"])))
    hosted.execute([pa.AdvectionRK4, Age, pa.DeleteParticle], dt=1800.0, runtime=case["runtime"])
    assert 0 < len(hosted) < 400 and sum(seen_errors) == 400 - len(hosted)
    assert np.array_equal(hosted.particle_id, fused.particle_id)
    for k in ("x", "y", "z", "t"):
        assert np.array_equal(hosted._data[k], fused._data[k]), k
    assert np.all(hosted.age == case["runtime"])
    assert hosted._last_stats["hosted"] and hosted._last_stats["launches"] == 2 * 12


# ---- tests/test_diffusion.py:81-126 of the reference: user-written kernels that draw random numbers -------------------------------
@pytest.mark.parametrize("lambd", [1, 5])
def test_randomexponential(gpu, fieldset, lambd):
    npart = 1000
    fieldset.add_context("lambd", lambd)
    np.random.seed(1234)
    pset = pa.ParticleSet(fieldset=fieldset, x=np.zeros(npart), y=np.zeros(npart), z=np.zeros(npart))

    def vertical_randomexponential(particles, fieldset):
        particles.z = np.random.exponential(scale=1 / fieldset.lambd, size=len(particles))

    pset.execute(vertical_randomexponential, runtime=np.timedelta64(1, "s"), dt=np.timedelta64(1, "s"))
    assert np.allclose(np.mean(pset.z), 1.0 / fieldset.lambd, rtol=0.1)


@pytest.mark.parametrize("mu", [0.8 * np.pi, np.pi])
@pytest.mark.parametrize("kappa", [2, 4])
def test_randomvonmises(gpu, fieldset, mu, kappa):
    import random

    npart = 10000
    fieldset.add_context("mu", mu)
    fieldset.add_context("kappa", kappa)
    random.seed(1234)
    AngleParticle = pa.Particle.add_variable(pa.Variable("angle"))
    pset = pa.ParticleSet(fieldset=fieldset, pclass=AngleParticle, x=np.zeros(npart), y=np.zeros(npart), z=np.zeros(npart))

    def vonmises(particles, fieldset):
        particles.angle = np.array([random.vonmisesvariate(fieldset.mu, fieldset.kappa) for _ in range(len(particles))])

    pset.execute(vonmises, runtime=np.timedelta64(1, "s"), dt=np.timedelta64(1, "s"))
    assert np.allclose(np.mean(pset.angle), mu, atol=0.1)
