"""The dedicated RK4 kernels of csrc/pk_fast_agrid.h (rectilinear A-grid, float64 coordinates) against the general
program of the same library -- t, state, ei, ids and the counters exactly, positions to FAST_VS_GENERAL_RTOL -- and against the
CPU oracle at 1e-12, on the inputs where the two code paths differ
most: sample points exactly on nodes and time levels (barycentric coordinate exactly 0: one level / plane is read),
particles of one wavefront on different time levels (the readfirstlane waterfall iterates), time-level rings, domain exits,
float32 fields / particles, missing depth axis, 3-D advection."""

from __future__ import annotations

import numpy as np
import pytest

from case_utils import build_fieldset, build_pset, compare, endtime_of, run_oracle

pytestmark = pytest.mark.gpu

# Until round 6 the dedicated kernel gave the BITS of the general program.  It now forms a barycentric coordinate as (x - a) * RN(1 / width)
# and u / (deg2m cos lat), v / deg2m, sum / 6 with reciprocals (pk_fast_agrid.h: PK_FAST_LEAN, within 1.5 ulp each, 41 instructions less per
# evaluation); the general program keeps the correctly rounded quotients.  Everything discrete still has to agree exactly; float32 particle
# storage rounds every step's position to float32, where a last-bit difference of the float64 sum can move the stored value by one float32 ulp.
FAST_VS_GENERAL_RTOL = 1e-12


def _scale(case):
    """Yardstick of the position comparisons: the coordinate scale (longitudes and metre axes run through 0: |x| alone is none there),
    |a - b| <= rtol * (|b| + scale) like tests/test_gpu_fuzz.py and the bench-size checks"""
    return float(max(np.abs(np.asarray(case["lon"])).max(), np.abs(np.asarray(case["lat"])).max()))


def _cmp_general(fast, gen, case, label):
    rtol = 5e-7 if case.get("spatial_dtype", "float64") == "float32" else FAST_VS_GENERAL_RTOL
    compare(fast, gen, rtol=rtol, atol_pos=rtol * _scale(case), check_state="all", label=label, skip=())


def _run(case, fast, nslots=None, endtime=None):
    import warnings

    import parcels_amd as pa

    fs = build_fieldset(case)
    fs.to_device(nslots=nslots)
    fs._engine.ctx.set_option("fast_path", 1 if fast else 0)
    pset = build_pset(case, fs)
    kernels = [getattr(pa.kernels, k) for k in case["kernels"]]
    kw = {"endtime": endtime_of(endtime)} if endtime is not None else {"runtime": float(case["runtime"])}
    err = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            pset.execute(kernels, dt=float(case["dt"]), **kw)
        except (pa.FieldOutOfBoundError, pa.FieldOutOfBoundSurfaceError, pa.FieldInterpolationError, pa.OutsideTimeInterval, pa.GeneralError) as e:
            err = type(e).__name__
    return {k: np.array(v) for k, v in pset._data.items()}, err, pset._last_stats


def _check(case, *, nslots=None, oracle=True, rtol=1e-12):
    fast, ferr, fstats = _run(case, True, nslots)
    gen, gerr, gstats = _run(case, False, nslots)
    assert ferr == gerr
    assert fstats["steps"] == gstats["steps"] and fstats["attempts"] == gstats["attempts"]
    _cmp_general(fast, gen, case, case["name"] + ": fast vs general")
    if oracle:
        ref, oerr, _ = run_oracle(case)
        assert ferr == oerr
        if ferr is None:
            compare(fast, ref, rtol=rtol, atol_pos=rtol * _scale(case), check_state="all", label=case["name"] + ": fast vs oracle", skip=())
    return fast


@pytest.mark.parametrize("kernel", ["AdvectionRK4", "AdvectionRK4_3D"])
@pytest.mark.parametrize("mesh", ["spherical", "flat"])
@pytest.mark.parametrize("fdt,sdt", [(np.float64, "float64"), (np.float32, "float64"), (np.float64, "float32"), (np.float32, "float32")])
def test_fast_equals_general_and_oracle(gpu, kernel, mesh, fdt, sdt):
    from oracle import cases

    case = cases.rect_agrid_case("fast_" + kernel, mesh=mesh, kernels=[kernel], seed=11, nx=40, ny=24, nz=7, nt=5, npart=3000, field_dtype=fdt,
                                 spatial_dtype=sdt, with_w=kernel.endswith("3D"), runtime=30 * 3600.0)
    _check(case, rtol=5e-7 if sdt == "float32" else 1e-12)


def test_points_on_nodes_and_time_levels(gpu):
    """x, y, z exactly on grid nodes, t exactly on level times, dt dividing the level spacing: every barycentric quotient is 0
    somewhere (div_by_recip's hardware-division branch) and lenT / lenZ switch between 1 and 2 inside one wavefront."""
    from oracle import cases

    case = cases.rect_agrid_case("fast_nodes", mesh="spherical", kernels=["AdvectionRK4"], seed=3, nx=37, ny=19, nz=6, nt=4, npart=4096,
                                 runtime=2 * 86400.0, dt=21600.0, level_dt=86400.0)
    lon, lat, depth = case["lon"], case["lat"], case["depth"]
    rng = np.random.default_rng(0)
    n = len(case["x"])
    on = rng.random(n) < 0.5
    case["x"] = np.where(on, lon[rng.integers(2, len(lon) - 2, n)], case["x"])
    case["y"] = np.where(rng.random(n) < 0.5, lat[rng.integers(2, len(lat) - 2, n)], case["y"])
    case["z"] = np.where(rng.random(n) < 0.5, depth[rng.integers(0, len(depth), n)], case["z"])  # incl. the surface (zeta == 0) and the bottom
    _check(case)


def test_staggered_release_times_share_a_wavefront(gpu):
    """Release times spread over all time levels, unsorted: lanes of one wavefront gather from different levels."""
    from oracle import cases

    case = cases.rect_agrid_case("fast_stagger", mesh="spherical", kernels=["AdvectionRK4"], seed=8, nx=30, ny=20, nz=5, nt=6, npart=5000,
                                 runtime=None, dt=3600.0, level_dt=43200.0)
    n = len(case["x"])
    case["t0"] = np.random.default_rng(1).uniform(0, 4 * 43200.0, n)
    case["t0"][::7] = 43200.0 * (np.arange(len(case["t0"][::7])) % 4)  # some exactly on a level
    case["endtime"] = 5 * 43200.0
    case["runtime"] = None
    fast, ferr, _ = _run(case, True, endtime=case["endtime"])
    gen, gerr, _ = _run(case, False, endtime=case["endtime"])
    assert ferr == gerr is None
    _cmp_general(fast, gen, case, "stagger")
    ref, oerr, _ = run_oracle(case, endtime=case["endtime"])
    compare(fast, ref, rtol=1e-12, atol_pos=1e-12 * _scale(case), check_state="all", label="stagger vs oracle", skip=())
    # and through a ring of 3 levels (pause / resume per launch window)
    ring, rerr, rstats = _run(case, True, nslots=3, endtime=case["endtime"])
    assert rerr is None and rstats["launches"] > 1
    compare(ring, fast, rtol=0.0, check_state="all", label="stagger ring", skip=())


@pytest.mark.parametrize("kernel", ["AdvectionRK4", "AdvectionRK4_3D"])
def test_domain_exits_and_backward_time(gpu, kernel):
    """Fast flow out of a small flat domain (right / left exits in x and y, surface and bottom in z) with the recovery kernels
    appended, then the same backwards in time."""
    from oracle import cases

    for sign in (1.0, -1.0):
        case = cases.rect_agrid_case("fast_exit", mesh="flat", kernels=[kernel, "DeleteParticle"], seed=21, nx=24, ny=16, nz=5, nt=4, npart=4000,
                                     with_w=kernel.endswith("3D"), vel=3.0, wscale=0.02, margin=0.02, runtime=36 * 3600.0, dt=sign * 3600.0)
        if sign < 0:
            case["t0"] = np.full(len(case["x"]), float(case["time_s"][-1]))
        fast, ferr, fst = _run(case, True)
        gen, gerr, gst = _run(case, False)
        assert ferr == gerr is None
        assert len(fast["x"]) < 4000, "nothing left the domain: the test does not test"
        _cmp_general(fast, gen, case, f"exit {sign}")
        ref, oerr, _ = run_oracle(case)
        compare(fast, ref, rtol=1e-12, atol_pos=1e-12 * _scale(case), check_state="all", label=f"exit vs oracle {sign}", skip=())


def test_errors_raise_the_same(gpu):
    """Without a recovery kernel the first out-of-bounds particle raises: same exception, same stop state."""
    from oracle import cases

    case = cases.rect_agrid_case("fast_raise", mesh="flat", kernels=["AdvectionRK4"], seed=22, nx=24, ny=16, nz=5, nt=4, npart=500, vel=3.0,
                                 margin=0.02, runtime=36 * 3600.0)
    fast, ferr, _ = _run(case, True)
    gen, gerr, _ = _run(case, False)
    assert ferr == gerr and ferr is not None
    _cmp_general(fast, gen, case, "raise")


def test_field_without_depth_axis(gpu):
    """2-D fields on a grid without a vertical axis (nz == 1 descriptors, lenZ never 2)."""
    from oracle import cases

    case = cases.rect_agrid_case("fast_2d", mesh="spherical", kernels=["AdvectionRK4"], seed=4, nx=33, ny=21, nz=1, nt=3, npart=2000, runtime=20 * 3600.0)
    if case["fields"]["U"].shape[1] == 1 and case.get("depth") is not None and len(case["depth"]) == 1:
        case["z"] = np.zeros(len(case["x"]))
    fast, ferr, _ = _run(case, True)
    gen, gerr, _ = _run(case, False)
    assert ferr == gerr
    _cmp_general(fast, gen, case, "2d")


def test_fast_path_is_taken_and_can_be_switched_off(gpu):
    """The kernel names the two settings launch differ: checked through the per-launch statistics being identical while the
    general program reports the staged LDS layout of the general path (smoke test of pk_set_option itself)."""
    import parcels_amd as pa
    from oracle import cases

    case = cases.rect_agrid_case("fast_opt", mesh="spherical", kernels=["AdvectionRK4"], seed=2, npart=256, runtime=6 * 3600.0)
    fs = build_fieldset(case)
    fs.to_device()
    with pytest.raises(pa._hip.HipLibraryError):
        fs._engine.ctx.set_option("no_such_option", 1)
    for v in (0, 1):
        fs._engine.ctx.set_option("fast_path", v)
        pset = build_pset(case, fs)
        pset.execute(pa.AdvectionRK4, dt=3600.0, runtime=6 * 3600.0)
        assert pset._last_stats["steps"] == 256 * 6
