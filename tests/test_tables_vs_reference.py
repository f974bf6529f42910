"""CPU: the small tables and filters of the hot path against the reference's REAL modules (under oracle/ref_shim.py): status codes, the
error classes and the order in which Kernel.execute raises them, the rows a ParticleFile writes at an output time
(particlefile.py:198-221), the Variables of a particle class that are written."""
import importlib

import numpy as np
import pytest

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")


def test_status_codes_and_error_order():
    import parcels_amd as pa
    import parcels_amd.statuscodes as mine

    m = ref_shim.load_reference()
    ref = m["statuscodes"]
    names = [k for k in dir(ref.StatusCode) if not k.startswith("_")]
    assert {k: int(getattr(ref.StatusCode, k)) for k in names} == {k: int(getattr(pa.StatusCode, k)) for k in names}
    assert sorted(k for k in dir(pa.StatusCode) if not k.startswith("_") and k[0].isupper()) == sorted(names)
    assert {c.__name__: v for c, v in ref.AllParcelsErrorCodes.items()} == {c.__name__: v for c, v in mine.AllParcelsErrorCodes.items()}
    for cls in ref.AllParcelsErrorCodes:  # same exception hierarchy (what user code catches)
        mcls = getattr(mine, cls.__name__)
        rb = [b.__name__ for b in cls.__mro__ if b.__module__.startswith("parcels")]
        mb = [b.__name__ for b in mcls.__mro__ if b.__module__.startswith("parcels_amd")]
        assert rb == mb, cls.__name__
    # the order in which the codes are checked after an iteration (kernel.py:239-245): the first present one raises
    assert [int(c) for c in m["kernel"].ErrorsToThrow] == [int(c) for c in mine.ErrorsToThrow]
    for code, func in m["kernel"].ErrorsToThrow.items():
        args = (np.array([1.0]),) if int(code) == 70 else (np.array([1.0]), np.array([2.0]), np.array([3.0]))
        with pytest.raises(Exception) as er:
            func(*args)
        with pytest.raises(Exception) as em:
            mine.ErrorsToThrow[int(code)](*args)
        assert type(er.value).__name__ == type(em.value).__name__, code


@pytest.mark.parametrize("seed", range(12))
def test_rows_written_at_an_output_time(seed):
    from parcels_amd.particlefile import _to_write_particles as mine

    ref = importlib.import_module("parcels._core.particlefile")._to_write_particles
    rng = np.random.default_rng(seed)
    n = 200
    dt = rng.choice([600.0, -600.0, 0.5, np.nan], size=n, p=[0.5, 0.3, 0.1, 0.1])
    t = np.round(rng.uniform(0, 7200, n) / 300) * 300
    t[rng.random(n) < 0.1] = np.nan
    pid = np.arange(n, dtype=np.int64)
    data = {"t": t, "dt": dt, "particle_id": pid}
    for tout in (0.0, 300.0, 3600.0, 7200.0, 7500.0):
        with np.errstate(invalid="ignore"):
            a = np.asarray(ref(data, tout))
            b = np.asarray(mine(data, tout))
        assert np.array_equal(a, b), (tout, a, b)


def test_variables_that_are_written():
    import parcels_amd as pa
    from parcels_amd.particlefile import _get_vars_to_write as mine

    m = ref_shim.load_reference()
    ref = importlib.import_module("parcels._core.particlefile")._get_vars_to_write
    RP = m["particle"]
    for spatial in (np.float32, np.float64):
        rc = RP.get_default_particle(spatial).add_variable([RP.Variable("age", dtype=np.float32, initial=0), RP.Variable("tmp", dtype=np.float64, initial=1, to_write=False),
                                                            RP.Variable("n", dtype=np.int32, initial=2)])
        mc = pa.get_default_particle(spatial).add_variable([pa.Variable("age", dtype=np.float32, initial=0), pa.Variable("tmp", dtype=np.float64, initial=1, to_write=False),
                                                            pa.Variable("n", dtype=np.int32, initial=2)])
        assert [(v.name, np.dtype(v.dtype), v.to_write) for v in ref(rc)] == [(v.name, np.dtype(v.dtype), v.to_write) for v in mine(mc)]


@pytest.mark.parametrize("kw", [dict(name=3), dict(name="2x"), dict(name="a b"), dict(name="class"), dict(name="ok", dtype="nonsense"),
                                dict(name="ok", to_write="once"), dict(name="ok", to_write=False, attrs={"units": "m"}),
                                dict(name="ok", dtype=np.int16, initial=4, attrs={"units": "m"})])
def test_variable_validation(kw):
    """Variable(...) refuses what the reference refuses, with the reference's exception type and message (particle.py:36-60)."""
    import parcels_amd as pa

    RP = ref_shim.load_reference()["particle"]
    res = []
    for V in (RP.Variable, pa.Variable):
        try:
            v = V(**kw)
            res.append(("ok", (v.name, np.dtype(v.dtype), v.initial, v.to_write, dict(v.attrs))))
        except (TypeError, ValueError) as e:
            res.append((type(e).__name__, str(e)))
    assert res[0] == res[1], res
