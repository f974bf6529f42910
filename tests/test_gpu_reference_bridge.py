"""GPU: parcels_amd.reference_bridge.HipBackend -- the binding of INTEGRATION.md section 2 -- end to end against the fixtures the
reference generated.  The reference tree is not on the GPU box: the backend gets attribute stand-ins of the reference's objects
(tests/bridge_utils.py), which tests/test_reference_bridge.py ties to the reference's real classes on the CPU; the particle columns are
a plain SoA dict with the reference's names and dtypes."""
from types import SimpleNamespace as NS

import numpy as np
import pytest

from bridge_utils import standin_fieldset
from case_utils import build_fieldset, build_pset, compare, load_golden, tolerance_for

pytestmark = pytest.mark.gpu


def _function(name):
    def f(particles, fieldset):  # the reference's kernels are plain functions; the bridge matches them by name
        raise AssertionError("built-in kernels run on the device")

    f.__name__ = name
    return f


class _Pset:
    """What Kernel.execute touches of the reference's ParticleSet: the SoA dict, len(), remove_indices()."""

    def __init__(self, data):
        self._data = data

    def __len__(self):
        return len(self._data["t"])

    def remove_indices(self, idx):
        self._data = {k: np.delete(v, idx, axis=0) for k, v in self._data.items()}


@pytest.mark.parametrize("name", ["agrid_sph_rk4_f64", "agrid_flat_rk4_3d_f64", "agrid_sph_rk4_f32part", "cgrid_curv_sph_rk4_3d_populated",
                                  "cgrid_rect_sph_rk4_3d", "agrid_sph_rk45", "diff_m1_constkh_flat", "agrid_flat_rk4_escape",
                                  "agrid_flat_rk4_3d_escape_delete"])
def test_backend_runs_reference_objects_like_the_reference(gpu, name):
    import parcels_amd as pa
    from parcels_amd.reference_bridge import HipBackend

    case, out, err = load_golden(name)
    assert case.get("t0") is None and not case.get("outputdt")  # one Kernel.execute from t = 0 to the end of the run
    ref_fs = standin_fieldset(case)
    kernels = list(case["kernels"])
    if "AdvectionRK45" in kernels:  # Kernel.__init__ of the reference (kernel.py:129-148) leaves these in the context
        sph = case["mesh"] == "spherical"
        ref_fs.context.update(RK45_tol=10 / (1852 * 60) if sph else 10, RK45_min_dt=1, RK45_max_dt=86400)
        ref_fs.context.update({k: v for k, v in (case.get("context") or {}).items()})
        if sph and "RK45_tol" in (case.get("context") or {}):
            ref_fs.context["RK45_tol"] = case["context"]["RK45_tol"] / (1852 * 60)
    backend = HipBackend(ref_fs, seed=int(case.get("seed", 0)))
    mine = build_pset(case, build_fieldset(case))  # only for its freshly initialised columns (names / dtypes of particle.py:182-222)
    if case.get("populate"):
        mine.populate_indices()
    pset = _Pset({k: np.array(v) for k, v in mine._data.items()})
    dt = float(case["dt"])
    pset._data["dt"][:] = dt  # ParticleSet.execute (particleset.py:419-423)
    endtime = float(case["runtime"]) if case.get("runtime") is not None else float(case["endtime"])
    funcs = [_function(k) for k in kernels]
    assert backend.supports(funcs)
    st = backend.execute(pset, funcs, endtime, dt)
    assert st["launches"] >= 1 and (st["reran"] > 0) == (err is not None)
    gone = pset._data["state"] == int(pa.StatusCode.Delete)  # Kernel.remove_deleted (kernel.py:98-106) is the caller's
    if gone.any():
        pset.remove_indices(np.flatnonzero(gone))
    compare(pset._data, out, rtol=tolerance_for(name, case), check_state="all", label="bridge " + name)
    if err is not None:  # the installed wrapper raises from these codes with the reference's own ErrorsToThrow table
        assert np.any(pset._data["state"] >= int(pa.StatusCode.Error))


def test_backend_declines_what_has_no_device_form(gpu):
    from parcels_amd.reference_bridge import HipBackend, UnsupportedByDevice, fieldset_from_reference

    case, _, _ = load_golden("agrid_sph_rk4_f64")
    ref_fs = standin_fieldset(case)
    backend = HipBackend(ref_fs)
    assert backend.supports([_function("AdvectionRK4")]) and not backend.supports([_function("AdvectionRK4"), _function("Ageing")])
    assert not backend.supports([_function("AdvectionRK4_3D")])  # no W field
    assert not backend.supports([_function("AdvectionRK45")])   # the RK45 context is missing
    ref_fs.fields["U"].interp_method = type("MyInterpolator", (), {})()
    with pytest.raises(UnsupportedByDevice, match="MyInterpolator"):
        fieldset_from_reference(ref_fs)
    ux = NS(fields={"U": NS(name="U", grid=NS(), data=None, interp_method=None)}, gridset=[], context={})
    with pytest.raises(UnsupportedByDevice, match="not a structured grid"):
        fieldset_from_reference(ux)


def Ageing(particles, fieldset):  # a kernel a user of the reference would write (module level: the translator reads its source)
    particles.age += particles.dt
    particles.state = np.where(particles.age > fieldset.max_age, 30, particles.state)  # StatusCode.Delete


def test_backend_compiles_the_users_kernels_into_the_launch(gpu):
    """A list with a user-written elementwise kernel goes to the GPU as ONE launch too (parcels_amd/jit.py): same result as
    parcels_amd's own ParticleSet running the same list, and the particles older than max_age are gone."""
    import parcels_amd as pa
    from parcels_amd.reference_bridge import HipBackend

    case, out, err = load_golden("agrid_sph_rk4_f64")
    ref_fs = standin_fieldset(case)
    ref_fs.context["max_age"] = 20 * 3600.0
    backend = HipBackend(ref_fs)
    fs = build_fieldset(case)
    fs.add_context("max_age", 20 * 3600.0)
    P = pa.get_default_particle(np.float64).add_variable(pa.Variable("age", dtype=np.float32, initial=0))
    mine = pa.ParticleSet(fs, pclass=P, x=np.asarray(case["x"]), y=np.asarray(case["y"]), z=case.get("z"), t=np.zeros(len(case["x"])))
    pset = _Pset({k: np.array(v) for k, v in mine._data.items()})
    pset._pclass = P  # (the reference's ParticleSet has it under the same name)
    dt, endtime = float(case["dt"]), float(case["runtime"])
    pset._data["dt"][:] = dt
    funcs = [_function("AdvectionRK4"), Ageing]
    assert backend.supports(funcs, pset), backend.jit_report
    st = backend.execute(pset, funcs, endtime, dt)
    assert st["launches"] == 1 and st["program"] == 100  # the dedicated A-grid kernel, the user kernel riding along
    mine.execute([pa.AdvectionRK4, Ageing], runtime=endtime, dt=dt)
    gone = pset._data["state"] == int(pa.StatusCode.Delete)
    assert gone.any() and not gone.all() or gone.all() or not gone.any()
    pset.remove_indices(np.flatnonzero(gone))
    for k in ("particle_id", "t", "x", "y", "z", "state", "age", "ei"):
        assert np.array_equal(pset._data[k], mine._data[k]), k
    assert np.all(pset._data["age"] <= 20 * 3600.0 + dt)
