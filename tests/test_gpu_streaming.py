"""Field-slab streaming (the WindowedArray replacement, _windowed_array.py:25-113) beyond one fresh forward run: a ring that is
re-used by later ParticleSets / other time directions, and fields of one FieldSet that live on different time axes."""

from __future__ import annotations

import numpy as np
import pytest

from case_utils import build_fieldset, build_pset, compare, endtime_of

pytestmark = pytest.mark.gpu


def _soa(pset):
    return {k: np.array(v) for k, v in pset._data.items()}


def test_one_windowed_fieldset_serves_forward_restart_and_backward_runs(gpu):
    """A forward run leaves the LAST levels in the ring; a second ParticleSet released at t = 0 on the same device FieldSet, and
    a backward run from the end, must find their own levels (stale slots are evicted, not mixed into the window)."""
    import parcels_amd as pa
    from oracle import cases

    case = cases.rect_agrid_case("ring_reuse", mesh="spherical", kernels=["AdvectionRK4"], seed=6, nt=9, npart=3000, dt=5000.0,
                                 runtime=7.5 * 86400.0, level_dt=86400.0)
    tend = float(case["time_s"][-1])
    ref_fs = build_fieldset(case)  # all levels resident
    win_fs = build_fieldset(case)
    win_fs.to_device(nslots=4)
    out = {}
    for tag, fs in (("ref", ref_fs), ("win", win_fs)):
        a = build_pset(case, fs)
        a.execute(pa.AdvectionRK4, dt=case["dt"], runtime=case["runtime"])
        b = build_pset(case, fs)  # second release at t = 0 after the ring moved to the end of the time axis
        b.execute(pa.AdvectionRK4, dt=case["dt"], runtime=2.5 * 86400.0)
        c = build_pset(dict(case, t0=np.full(len(case["x"]), tend)), fs)
        c.execute(pa.AdvectionRK4, dt=-case["dt"], endtime=endtime_of(tend - 6.2 * 86400.0))
        # particles far apart in time inside ONE set: early ones finish (deleted by endtime semantics = EndofLoop), late ones start later
        t0 = np.where(np.arange(len(case["x"])) % 2 == 0, 0.0, 6.0 * 86400.0)
        d = build_pset(dict(case, t0=t0), fs)
        d.execute(pa.AdvectionRK4, dt=case["dt"], endtime=endtime_of(7.8 * 86400.0))
        out[tag] = [_soa(p) for p in (a, b, c, d)]
        if tag == "win":
            assert all(p._last_stats["launches"] > 1 for p in (a, c, d)), "the ring was not exercised"
    for k, (w, r) in enumerate(zip(out["win"], out["ref"])):
        compare(w, r, rtol=0.0, check_state="all", label=f"run {k}", skip=())


def test_streamed_scalar_fields_on_their_own_time_axis(gpu):
    """U, V on 3 levels four days apart (resident), Kh_zonal / Kh_meridional on 9 daily levels streamed through a ring of 4:
    every field is planned on its own axis and the particles pause where the intersection of the windows ends."""
    import parcels_amd as pa

    rng = np.random.default_rng(12)
    nx, ny = 40, 30
    lon = np.linspace(-2.0e4, 2.0e4, nx)
    lat = np.linspace(-1.5e4, 1.5e4, ny)
    depth = np.array([0.0, 100.0])

    def smooth(shape, scale):
        a = rng.standard_normal(shape)
        for ax in (2, 3):
            a = (a + np.roll(a, 1, ax) + np.roll(a, -1, ax)) / 3
        return scale * a

    md = pa.SGrid2DMetadata(
        node_dimensions=("XG", "YG"), node_coordinates=("lon", "lat"),
        face_dimensions=(pa.FaceNodePadding("XC", "XG", pa.Padding.LOW), pa.FaceNodePadding("YC", "YG", pa.Padding.LOW)),
        vertical_dimensions=(pa.FaceNodePadding("ZC", "depth", pa.Padding.BOTH),))
    dims = ("time", "depth", "YG", "XG")

    def coords(time_s):
        return {"lon": (("XG",), lon), "lat": (("YG",), lat), "depth": (("depth",), depth), "time": (("time",), np.asarray(time_s, dtype=np.float64))}

    t_uv = np.arange(3) * 4 * 86400.0
    t_kh = np.arange(9) * 86400.0
    uv = {"U": (dims, smooth((3, 2, ny, nx), 0.05)), "V": (dims, smooth((3, 2, ny, nx), 0.05))}
    xs, ys = lon / lon[-1], lat / lat[-1]
    tz = (1 + 0.1 * np.arange(9))[:, None, None, None]
    kh = {"Kh_zonal": (dims, 2.0 * tz * (1 + 0.5 * np.tanh(3 * xs))[None, None, None, :] * np.ones((9, 2, ny, nx))),
          "Kh_meridional": (dims, 2.0 * tz * (1 + 0.3 * np.tanh(2 * ys))[None, None, :, None] * np.ones((9, 2, ny, nx)))}
    n = 4000
    x = rng.uniform(lon[0] * 0.3, lon[-1] * 0.3, n)
    y = rng.uniform(lat[0] * 0.3, lat[-1] * 0.3, n)
    res = {}
    for ns in (None, 4):
        m_uv = pa.FieldSet.from_sgrid_conventions(pa.Dataset(uv, coords(t_uv), sgrid=md), mesh="flat").models
        m_kh = pa.FieldSet.from_sgrid_conventions(pa.Dataset(kh, coords(t_kh), sgrid=md), mesh="flat", vector_fields={}).models
        fs = pa.FieldSet(m_uv + m_kh)
        for f in ("Kh_zonal", "Kh_meridional"):
            fs.fields[f].interp_method = pa.XLinear()
        fs.add_context("dres", 100.0)
        fs.to_device(nslots=ns)
        eng = fs._engine
        if ns is not None:
            assert eng.field_nslots["U"] == 3 and eng.field_nslots["Kh_zonal"] == 4  # U resident, Kh streamed
        pset = pa.ParticleSet(fs, pclass=pa.get_default_particle(np.float64), x=x, y=y, z=np.full(n, 50.0), t=np.zeros(n), seed=7)
        pset.execute([pa.AdvectionDiffusionM1, pa.kernels.DeleteParticle], dt=3000.0, runtime=7.6 * 86400.0)
        res[ns] = _soa(pset)
        if ns is not None:
            assert pset._last_stats["launches"] > 1, "the ring was not exercised"
    compare(res[4], res[None], rtol=0.0, check_state="all", label="own time axes", skip=())


def test_field_eval_at_explicit_points_through_a_ring(gpu):
    """Field.eval / VectorField.eval (field.py:145-195, 250-304) with the levels streaming through a ring of 3: the points are
    served one resident window at a time and the values equal those of a fully resident FieldSet, bit for bit."""
    from oracle import cases

    case = cases.rect_agrid_case("ring_eval", mesh="spherical", kernels=["AdvectionRK4"], seed=9, nt=8, npart=2000, with_w=True, level_dt=86400.0)
    rng = np.random.default_rng(3)
    n = len(case["x"])
    t = rng.uniform(-1000.0, 7 * 86400.0 + 1000.0, n)  # unsorted, a few outside the time interval
    t[::50] = 86400.0 * rng.integers(0, 8, len(t[::50]))  # exactly on levels
    res = {}
    for ns in (None, 3):
        fs = build_fieldset(case)
        fs.to_device(nslots=ns)
        uvw = fs.UVW.eval(t, case["z"], case["y"], case["x"])
        usc = fs.U.eval(t, case["z"], case["y"], case["x"])
        res[ns] = (np.array(uvw[0]), np.array(uvw[1]), np.array(uvw[2]), np.array(usc), np.array(fs._engine.last_sample_state))
    for a, b in zip(res[3], res[None]):
        assert np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b)
    assert np.any(res[None][4] == 70) and np.any(res[None][0] != 0)


def test_to_windowed_arrays_is_the_three_slot_ring(gpu):
    """FieldSet.to_windowed_arrays() (fieldset.py:142-173): the reference's opt-in rolling window -- here the device ring, requested before
    any device exists -- leaves the trajectories of a fully resident FieldSet, bit for bit, and describe() reports the ring."""
    import io

    import parcels_amd as pa
    from oracle import cases

    case = cases.rect_agrid_case("ring_api", mesh="spherical", kernels=["AdvectionRK4"], seed=11, nt=8, npart=1500, level_dt=86400.0)
    res = {}
    for windowed in (False, True):
        fs = build_fieldset(case)
        if windowed:
            assert fs.to_windowed_arrays() is fs
        pset = pa.ParticleSet(fs, pclass=pa.get_default_particle(np.float64), x=case["x"], y=case["y"], z=case["z"], t=np.zeros(len(case["x"])))
        pset.execute([pa.AdvectionRK4, pa.kernels.DeleteParticle], dt=3600.0, runtime=6.5 * 86400.0)
        res[windowed] = _soa(pset)
        buf = io.StringIO()
        fs.describe(buf)
        if windowed:
            assert fs._engine.field_nslots["U"] == 3 and pset._last_stats["launches"] > 1 and "ring of 3 of 8 levels" in buf.getvalue()
        else:
            assert "all 8 levels resident" in buf.getvalue()
    compare(res[True], res[False], rtol=0.0, check_state="all", label="to_windowed_arrays", skip=())


@pytest.mark.parametrize("kernels", [["AdvectionRK4"], ["AdvectionEE", "DeleteParticle"], ["AdvectionRK45"]], ids=lambda k: "+".join(k))
@pytest.mark.parametrize("nslots", [None, 3, 4])  # (a ring of 2 cannot hold a step that straddles a level: these releases are staggered by quarter steps)
def test_call_wide_time_error_through_a_level_ring(gpu, kernels, nslots):
    """The reference's call-wide OutsideTimeInterval (field.py:31-44) in a call that takes SEVERAL launches: six 2-hour levels stream
    through a ring of 3 / 4 slots, staggered releases run one step past the last level.  The sample that fails shows up in the LAST
    launch of a pass; the call restarts from the device checkpoint with it listed (DeviceEngine.execute) -- same columns, same
    exception, same survivors as the oracle's batch loop (pinned to the reference on this class by the twe_* fixtures), for every ring."""
    from case_utils import compare, run_hip, run_oracle
    from oracle import cases

    case = cases.rect_agrid_case("twe_ring", mesh="spherical", kernels=kernels, seed=77, nt=6, level_dt=7200.0, stagger=True, npart=200, dt=1800.0)
    case["t0"] = np.asarray(case["t0"]) * 0.5  # (stagger in units of the half step of THIS dt)
    tl = float(case["time_s"][-1])
    case["runtime"] = tl + (2.0 if "AdvectionRK45" in kernels else 1.0) * 1800.0 - float(np.min(case["t0"]))
    if "AdvectionRK45" in kernels:
        case["context"] = {"RK45_tol": 500.0, "RK45_min_dt": 10.0, "RK45_max_dt": 3600.0}
    ref, oerr, _ = run_oracle(case)
    got, gerr, st = run_hip(case, nslots=nslots)
    assert gerr == oerr
    assert st["reran"] >= 1 and len(st["time_error_keys"]) >= 1
    if nslots is not None:
        assert st["launches"] > 2  # streamed: several launches per pass, and at least one repeated pass
    compare(got, ref, rtol=1e-11, check_state="all", label=f"ring {nslots} {kernels}", skip=("dt",) if "AdvectionRK45" in kernels else ())
