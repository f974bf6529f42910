"""CPU: what the host API refuses, against the reference's REAL classes (under oracle/ref_shim.py): the same wrong calls to
ParticleClass.add_variable, Kernel(...) and ParticleSet(...) raise the same exception type -- and the same message where the message
does not print an object's repr."""
import numpy as np
import pytest

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")


def _outcome(f):
    try:
        f()
        return ("ok", None)
    except Exception as e:
        return (type(e).__name__, str(e))


def _fieldsets():
    from case_utils import build_fieldset
    from oracle import cases
    from oracle.make_golden import build_ref_fieldset

    case = cases.rect_agrid_case("val", mesh="flat", kernels=["AdvectionRK4"], seed=1, npart=4, nx=6, ny=5, nz=2, nt=2)
    return build_ref_fieldset(case)[0], build_fieldset(case)


def _same(a, b, compare_message=True, same_type=True):
    if same_type:
        assert a[0] == b[0], (a, b)
    else:
        assert (a[0] == "ok") == (b[0] == "ok"), (a, b)
    if compare_message and a[0] != "ok" and "object at 0x" not in a[1] and "<function" not in a[1]:
        assert a[1] == b[1], (a, b)


@pytest.mark.parametrize("which", ["not_a_variable", "list_ok", "core_name"])
def test_add_variable(which):
    import parcels_amd as pa

    RP = ref_shim.load_reference()["particle"]
    out = []
    for mod, P in ((RP, RP.get_default_particle(np.float32)), (pa, pa.get_default_particle(np.float32))):
        V = mod.Variable
        arg = {"not_a_variable": ["a"], "list_ok": [V("a"), V("b", dtype=np.float64)], "core_name": V("x")}[which]
        o = _outcome(lambda: P.add_variable(arg))
        out.append(o)
    _same(out[0], out[1], compare_message=False)


def test_add_variable_duplicate_in_one_call_is_refused_here():
    """The reference checks new names against the existing ones only (particle.py:116-121), so two new Variables of one name pass
    and shadow each other in the data dict; here a name maps to one device column and the second is refused."""
    import parcels_amd as pa

    RP = ref_shim.load_reference()["particle"]
    assert _outcome(lambda: RP.get_default_particle(np.float32).add_variable([RP.Variable("a"), RP.Variable("a")]))[0] == "ok"
    assert _outcome(lambda: pa.get_default_particle(np.float32).add_variable([pa.Variable("a"), pa.Variable("a")]))[0] == "ValueError"


def Good(particles, fieldset):
    particles.dx += 1


def BadSignature(p, f, extra):
    pass


def BadNames(a, b):
    pass


@pytest.mark.parametrize("which", ["not_a_list", "empty", "not_a_function", "signature", "names", "rk45_without_next_dt", "ok"])
def test_kernel_construction(which):
    import parcels_amd as pa
    from parcels_amd.kernel import Kernel as MyKernel

    m = ref_shim.load_reference()
    ref_fs, my_fs = _fieldsets()
    rset = m["particleset"].ParticleSet(ref_fs, x=[1.0], y=[1.0])
    mset = pa.ParticleSet(my_fs, x=[1.0], y=[1.0])
    rk, mk = m["kernels"], pa
    args = {
        "not_a_list": (lambda k: Good),
        "empty": (lambda k: []),
        "not_a_function": (lambda k: [3]),
        "signature": (lambda k: [BadSignature]),
        "names": (lambda k: [BadNames]),
        "rk45_without_next_dt": (lambda k: [k.AdvectionRK45]),
        "ok": (lambda k: [k.AdvectionRK4, Good]),
    }[which]
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = _outcome(lambda: m["kernel"].Kernel(args(rk), rset))
        b = _outcome(lambda: MyKernel(args(mk), mset))
    _same(a, b, compare_message=which in ("empty", "rk45_without_next_dt"))


@pytest.mark.parametrize("which", ["length_mismatch", "z_length3", "t_length", "var_length", "unknown_kwarg", "ids_length", "scalar_ok"])
def test_particleset_construction(which):
    import parcels_amd as pa

    m = ref_shim.load_reference()
    ref_fs, my_fs = _fieldsets()
    kw = {
        "length_mismatch": dict(x=[1.0, 2.0], y=[1.0]),
        "z_length3": dict(x=[1.0, 2.0, 3.0], y=[1.0, 2.0, 3.0], z=[0.0, 1.0]),  # (a single depth is broadcast here, an extension)
        "t_length": dict(x=[1.0, 2.0], y=[1.0, 2.0], t=None),
        "var_length": dict(x=[1.0, 2.0], y=[1.0, 2.0], age=[1.0]),
        "unknown_kwarg": dict(x=[1.0], y=[1.0], nonsense=[1.0]),
        "ids_length": dict(x=[1.0, 2.0], y=[1.0, 2.0], particle_ids=[5]),
        "scalar_ok": dict(x=1.0, y=2.0),
    }[which]
    RP = m["particle"]
    rclass = RP.get_default_particle(np.float32).add_variable([RP.Variable("age", dtype=np.float32, initial=0)])
    mclass = pa.get_default_particle(np.float32).add_variable([pa.Variable("age", dtype=np.float32, initial=0)])
    rkw, mkw = dict(kw), dict(kw)
    if which == "t_length":
        rkw["t"] = np.array([0], dtype="timedelta64[s]")
        mkw["t"] = np.array([0.0])
    a = _outcome(lambda: m["particleset"].ParticleSet(ref_fs, pclass=rclass, **rkw))
    b = _outcome(lambda: pa.ParticleSet(my_fs, pclass=mclass, **mkw))
    _same(a, b, compare_message=False, same_type=which != "ids_length")  # the reference trips over list.shape there


def test_data_indices_and_remove_deleted_like_the_reference():
    import parcels_amd as pa
    from parcels_amd.kernel import Kernel as MyKernel

    m = ref_shim.load_reference()
    ref_fs, my_fs = _fieldsets()
    x = np.linspace(0.5, 3.0, 8)
    rset = m["particleset"].ParticleSet(ref_fs, x=x, y=np.ones(8))
    mset = pa.ParticleSet(my_fs, x=x, y=np.ones(8))
    for s in (rset, mset):
        s._data["state"][:] = [10, 30, 10, 60, 30, 0, 10, 41]
    for args in (("state", 30), ("state", [30, 60]), ("state", np.array([10])), ("particle_id", [0, 7, 99]), ("state", 10, True)):
        assert np.array_equal(rset.data_indices(*args), mset.data_indices(*args)), args
    m["kernel"].Kernel([m["kernels"].AdvectionRK4], rset).remove_deleted(rset)
    MyKernel([pa.AdvectionRK4], mset).remove_deleted(mset)
    assert len(rset) == len(mset) == 6
    for k in ("particle_id", "state", "x"):
        assert np.array_equal(rset._data[k], mset._data[k])
    a = _outcome(lambda: m["kernel"].Kernel([m["kernels"].AdvectionRK4], rset).merge(3))
    b = _outcome(lambda: MyKernel([pa.AdvectionRK4], mset).merge(3))
    assert a[0] == b[0] == "TypeError"
    a = _outcome(lambda: m["particleset"].ParticleSet.from_particlefile(ref_fs, None, "x"))
    b = _outcome(lambda: pa.ParticleSet.from_particlefile(my_fs, None, "x"))
    assert a == b and a[0] == "NotImplementedError"


def test_kernel_merge_and_write_status():
    """Kernel.merge raises a TypeError about its own constructor in the reference (kernel.py:168-172 passes three arguments to a two-argument
    __init__) and set_variable_write_status an AttributeError (an ndarray has no such method): here both do what their docstrings say."""
    import parcels_amd as pa
    from parcels_amd.kernel import Kernel as MyKernel

    _, my_fs = _fieldsets()
    P = pa.get_default_particle(np.float32).add_variable([pa.Variable("age", dtype=np.float32, initial=0)])
    pset = pa.ParticleSet(my_fs, pclass=P, x=[1.0], y=[1.0])
    k = MyKernel([pa.AdvectionRK4], pset).merge(MyKernel([Good], pset))
    assert [f.__name__ for f in k._kernels] == ["AdvectionRK4", "Good"] and k.fieldset is my_fs and k.pclass is P
    other = pa.ParticleSet(my_fs, x=[1.0], y=[1.0])
    with pytest.raises(AssertionError, match="different particle types"):
        MyKernel([pa.AdvectionRK4], pset).merge(MyKernel([pa.AdvectionRK4], other))
    assert [v.to_write for v in pset._pclass.variables if v.name == "age"] == [True]
    pset.set_variable_write_status("age", False)
    assert [v.to_write for v in pset._pclass.variables if v.name == "age"] == [False]
    assert [v.to_write for v in P.variables if v.name == "age"] == [True]  # the shared class is untouched
    with pytest.raises(KeyError):
        pset.set_variable_write_status("nope", False)


@pytest.mark.parametrize("which", ["dt_zero", "dt_type", "dt_none", "runtime_negative", "runtime_type", "runtime_and_endtime", "neither", "endtime_type",
                                   "kernel_not_a_function", "endtime_outside_fields"])
def test_execute_refuses_what_the_reference_refuses(which):
    """ParticleSet.execute's checks of dt / runtime / endtime (particleset.py:405-420, 473-560) run before anything touches a device: the
    same wrong calls raise the same exception -- same type, same message -- from the reference's real ParticleSet and from this one."""
    import warnings

    import parcels_amd as pa

    m = ref_shim.load_reference()
    ref_fs, my_fs = _fieldsets()
    left = ref_fs.time_interval.left
    s = lambda v: np.timedelta64(int(v), "s")  # noqa: E731
    kw = {
        "dt_zero": dict(dt=s(0), runtime=s(10)),
        "dt_type": dict(dt="fast", runtime=s(10)),
        "dt_none": dict(dt=None, runtime=s(10)),
        "runtime_negative": dict(dt=s(1), runtime=s(-5)),
        "runtime_type": dict(dt=s(1), runtime="long"),
        "runtime_and_endtime": dict(dt=s(1), runtime=s(5), endtime=left + s(5)),
        "neither": dict(dt=s(1)),
        "endtime_type": dict(dt=s(1), endtime=5.0),
        "kernel_not_a_function": dict(dt=s(1), runtime=s(5)),
        "endtime_outside_fields": dict(dt=s(1), endtime=left + s(10**9)),
    }[which]
    out = []
    for mod, fs, kernels in ((m, ref_fs, m["kernels"]), (pa, my_fs, pa)):
        PS = mod["particleset"].ParticleSet if isinstance(mod, dict) else mod.ParticleSet
        t = [left + s(10), left + s(20)] if isinstance(mod, dict) else [10.0, 20.0]
        pset = PS(fs, x=[1.0, 2.0], y=[1.0, 1.0], t=np.array(t))
        kern = 3 if which == "kernel_not_a_function" else kernels.AdvectionRK4
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out.append(_outcome(lambda: pset.execute(kern, **kw)))  # noqa: B023
    assert out[0][0] != "ok", out
    if out[0] == ("TypeError", "__repr__ returned non-string (type _Any)"):  # the message prints a TimeInterval, whose repr module is a stub here
        assert out[1][0] == "ValueError", out
        return
    _same(out[0], out[1], compare_message="object at 0x" not in out[0][1])
