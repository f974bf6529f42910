"""CPU: ParticleSet construction against the reference's REAL ParticleSet (src/parcels/_core/particleset.py + particle.py, loaded
unmodified under oracle/ref_shim.py): for the same positions, release times, particle classes (float32 / float64 storage, user Variables
with scalar and per-particle initial values) both build the same SoA dict -- same columns in the same order, same dtypes, same values."""
import numpy as np
import pytest

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")


def _fieldsets():
    from case_utils import build_fieldset
    from oracle import cases
    from oracle.make_golden import build_ref_fieldset

    case = cases.rect_agrid_case("pset", mesh="spherical", kernels=["AdvectionRK4"], seed=1, npart=4, nx=6, ny=5, nz=3, nt=3)
    return build_ref_fieldset(case)[0], build_fieldset(case)


@pytest.mark.parametrize("spatial", [np.float32, np.float64])
@pytest.mark.parametrize("with_t", [False, True])
@pytest.mark.parametrize("with_z", [False, True])
@pytest.mark.parametrize("nvars", [0, 3])
def test_same_particle_columns(spatial, with_t, with_z, nvars):
    import parcels_amd as pa

    ref_fs, my_fs = _fieldsets()
    m = ref_shim.load_reference()
    RP = m["particle"]
    n = 9
    rng = np.random.default_rng(3)
    x, y = rng.uniform(0, 300, n), rng.uniform(-70, 70, n)
    z = rng.uniform(0, 4000, n) if with_z else None
    tsec = np.round(rng.uniform(0, 86400, n)) if with_t else None
    spec = [("age", np.float32, 0), ("temp", np.float64, 12.5), ("tag", np.int32, 7)][:nvars]
    rclass = RP.get_default_particle(spatial)
    mclass = pa.get_default_particle(spatial)
    if spec:
        rclass = rclass.add_variable([RP.Variable(nm, dtype=dt, initial=init) for nm, dt, init in spec])
        mclass = mclass.add_variable([pa.Variable(nm, dtype=dt, initial=init) for nm, dt, init in spec])
    rkw, mkw = {}, {}
    if with_z:
        rkw["z"] = mkw["z"] = z
    if with_t:
        rkw["t"] = (tsec * 1e9).round().astype("int64").astype("timedelta64[ns]")
        mkw["t"] = tsec
    if spec:  # a per-particle initial value for one of the user Variables
        rkw["temp"] = mkw["temp"] = np.linspace(0, 1, n)
    rset = m["particleset"].ParticleSet(ref_fs, pclass=rclass, x=x, y=y, **rkw)
    mset = pa.ParticleSet(my_fs, pclass=mclass, x=x, y=y, **mkw)
    a, b = rset._data, mset._data
    assert list(a) == list(b), (list(a), list(b))
    for k in a:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (k, a[k].dtype, b[k].dtype, a[k].shape, b[k].shape)
        assert np.array_equal(a[k], b[k], equal_nan=True), (k, a[k], b[k])
    assert len(rset) == len(mset) == n
    assert [v.name for v in rclass.variables] == [v.name for v in mclass.variables]
    assert [np.dtype(v.dtype) for v in rclass.variables] == [np.dtype(v.dtype) for v in mclass.variables]
    assert [v.to_write for v in rclass.variables] == [v.to_write for v in mclass.variables]


def test_len_of_selections_like_the_reference():
    """particlesetview.py:83-84: `len(view)` is `len(view._index)` -- and a kernel's `particles` is a boolean mask over the WHOLE set, as is every
    selection made from it (by mask or by integer indices, :44-74): their len() is the size of the whole set; `len(view.x)` is the number of
    rows.  HostParticles, the `particles` of the host path, answers the same."""
    import parcels_amd as pa
    from parcels_amd.hostkernels import HostParticles

    m = ref_shim.load_reference()
    View = m["particlesetview"].ParticleSetView
    P = pa.get_default_particle(np.float32)
    n = 9
    data = {v.name: np.zeros(n, dtype=v.dtype) for v in P.variables}
    data["x"][:] = np.arange(n)
    data["particle_id"][:] = np.arange(n)
    ev = np.arange(n) % 3 != 0
    a, b = View(data, ev.copy(), P), HostParticles(data, np.flatnonzero(ev), by_mask=True)
    assert len(a) == len(b) == n and len(a.x) == len(b.x) == ev.sum()
    big = np.asarray(a.x) > 4
    assert len(a[big]) == len(b[np.asarray(b.x) > 4]) == n and len(a[big].x) == len(b[np.asarray(b.x) > 4].x) == 3
    idx = np.where(big)
    assert len(a[idx]) == len(b[idx]) == n and len(a[idx].x) == len(b[idx].x) == 3
    whole = np.asarray(data["x"]) < 2  # a mask over the whole set, on a selection: selects from the whole set
    assert list(np.asarray(a[whole].x)) == list(np.asarray(b[whole].x)) == [0.0, 1.0]
