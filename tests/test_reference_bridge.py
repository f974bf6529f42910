"""parcels_amd.reference_bridge against the reference's REAL classes (CPU; skipped where /root/reference is absent).

The bridge is the binding INTEGRATION.md section 2 describes, as running code: it reads the reference's FieldSet / XGrid / Field /
VectorField / ParticleSet by attribute.  Here (1) every kind of reference FieldSet the fixtures use converts to exactly the
parcels_amd.FieldSet the tests build directly from the same case -- and so do the attribute stand-ins the GPU tests use in place of the
reference; (2) `install` on the reference's own kernel module sends `ParticleSet.execute(AdvectionRK4, ...)` of the reference's own
ParticleSet to the backend with the reference's SoA dict, and leaves user-written kernels to the NumPy loop."""
import numpy as np
import pytest

from bridge_utils import assert_same_fieldset, standin_fieldset
from case_utils import build_fieldset
from oracle import cases, ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")


def _cases():
    c = [cases.rect_agrid_case("b_sph", mesh="spherical", kernels=["AdvectionRK4"], seed=1, npart=20),
         cases.rect_agrid_case("b_flat3d", mesh="flat", kernels=["AdvectionRK4_3D"], seed=2, npart=20, with_w=True, field_dtype=np.float32),
         cases.curv_cgrid_case("b_curv", mesh="spherical", kernels=["AdvectionRK4_3D"], seed=3)]
    d = dict(cases.rect_agrid_case("b_const", mesh="flat", kernels=["AdvectionDiffusionM1"], seed=4, npart=20))
    d["constants"] = {"Kh_zonal": 12.5, "Kh_meridional": 3.0}
    d["context"] = {"dres": 0.01}
    c.append(d)
    return c


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_reference_fieldsets_convert_to_the_fieldset_of_the_case(case):
    from oracle.make_golden import build_ref_fieldset
    from parcels_amd.reference_bridge import fieldset_from_reference

    ref_fs, _ = build_ref_fieldset(case)
    mine = build_fieldset(case)
    assert_same_fieldset(fieldset_from_reference(ref_fs), mine)
    assert_same_fieldset(fieldset_from_reference(standin_fieldset(case)), mine)  # what the GPU tests feed the bridge


class _RecordingEngine:
    """Stands in for the device: records what reaches it and finishes the batch (every particle at endtime, EndofLoop)."""

    def __init__(self):
        self.calls = []
        self.device_variables = []

    def bind_particles(self, data):
        self.data = data
        for k, dt in (("t", np.float64), ("dt", np.float64), ("state", np.int32), ("ei", np.int32), ("particle_id", np.int64)):
            assert data[k].dtype == dt, (k, data[k].dtype)  # pk_particles_desc's column types (include/parcels_hip.h)
        assert data["ei"].ndim == 2 and all(data[k].dtype in (np.float32, np.float64) for k in ("x", "y", "z", "dx", "dy", "dz"))

    def h2d(self):
        pass

    def set_user_program(self, program):
        assert program is None  # built-in lists carry no compiled user kernels

    def d2h(self):
        pass

    def execute(self, ids, *, endtime, dt0, context, seed, have_guess0, sort_by_cell, t_start, in_place_variables=False):
        self.calls.append(dict(ids=list(ids), endtime=endtime, dt0=dt0, t_start=t_start, n=len(self.data["t"])))
        self.data["t"][:] = endtime
        self.data["state"][:] = 2  # StatusCode.EndofLoop
        return {"steps": 0, "state_counts": {2: len(self.data["t"])}}


def test_installed_backend_takes_builtin_kernel_lists_and_leaves_user_kernels(monkeypatch):
    from oracle.make_golden import build_ref_fieldset
    from parcels_amd import reference_bridge as rb  # (4 below: PK_KERNEL_ADVECTION_RK4 of include/parcels_hip.h)

    m = ref_shim.load_reference()
    case = cases.rect_agrid_case("b_dispatch", mesh="spherical", kernels=["AdvectionRK4"], seed=5, npart=12, dt=1800.0, runtime=4 * 1800.0)
    ref_fs, _ = build_ref_fieldset(case)
    eng = _RecordingEngine()
    monkeypatch.setattr(rb.HipBackend, "engine", property(lambda self: eng))
    undo = rb.install(m["kernel"])
    try:
        pset = m["particleset"].ParticleSet(ref_fs, pclass=m["particle"].get_default_particle(np.float64), x=case["x"], y=case["y"], z=case["z"])
        pset.execute(m["kernels"].AdvectionRK4, runtime=np.timedelta64(7200, "s"), dt=np.timedelta64(1800, "s"), verbose_progress=False)
        assert eng.calls == [dict(ids=[4], endtime=7200.0, dt0=1800.0, t_start=0.0, n=12)]
        assert np.all(pset._data["t"] == 7200.0) and eng.data is pset._data  # the reference's own SoA dict crossed the boundary
        assert isinstance(ref_fs._hip_backend, rb.HipBackend)  # one backend (one device copy) per FieldSet

        def Ageing(particles, fieldset):  # a user-written kernel: no device form, the reference's NumPy loop runs it
            particles.dx += 0.0

        eng.calls.clear()
        pset2 = m["particleset"].ParticleSet(ref_fs, pclass=m["particle"].get_default_particle(np.float64), x=case["x"], y=case["y"], z=case["z"])
        pset2.execute([m["kernels"].AdvectionRK4, Ageing], runtime=np.timedelta64(3600, "s"), dt=np.timedelta64(1800, "s"), verbose_progress=False)
        assert eng.calls == [] and np.all(pset2._data["t"] == 3600.0)
        assert not np.array_equal(pset2._data["x"], np.asarray(case["x"]))  # ... and really advected
    finally:
        undo()
    assert m["kernel"].Kernel.execute.__qualname__.startswith("Kernel.")
