"""The dedicated RK4 kernels of csrc/pk_fast_cgrid.h (CGrid_Velocity on a spherical curvilinear C-grid, BASELINE configs 3 / 4)
against the general program of the same library -- t, state, ei, ids and the step / attempt counters exactly, positions to
FAST_VS_GENERAL_RTOL -- and against the CPU oracle at 1e-12, on the inputs where the two code paths differ most: cell crossings in every stage (neighbour probe + record / field fetch),
particles leaving the mesh (table walk, GridSearchingError, DeleteParticle), meshes where neighbour probing is off, level rings
with several launches, particles of one wavefront on different time levels, backward time, float32 particles and float64 fields."""

from __future__ import annotations

import numpy as np
import pytest

from case_utils import build_fieldset, build_pset, compare, endtime_of, run_oracle

pytestmark = pytest.mark.gpu

PROGRAM_FAST_CGRID = 101
# Until round 6 the dedicated kernels gave the BITS of the general program.  They now form the sines / cosines of a sample point from
# those of the particle's own position (pk_fast_cgrid.h: sincos_near -- absolute error <= 1.2e-16, what a libm gives, in 13 operations
# instead of 50), and the general program, which knows no "own position", keeps its full-range routine: last-bit differences in the
# velocity, 1e-15 .. 7e-13 of the coordinate after tens of steps.  Everything discrete still has to agree exactly.
FAST_VS_GENERAL_RTOL = 1e-12


def _scale(case):
    return float(max(np.abs(case["lon"]).max(), np.abs(case["lat"]).max()))


def _run(case, fast, nslots=None, endtime=None, probe=None, sort=False, pairs=False):
    import warnings

    import parcels_amd as pa

    from parcels_amd.engine import DeviceEngine

    fs = build_fieldset(case)
    fs.__dict__["_engine"] = DeviceEngine(fs, nslots=nslots, neighbour_probe=probe or 0)  # what FieldSet.to_device does, plus the probe switch
    fs._engine.ctx.set_option("fast_cgrid", 1 if fast else 0)
    # the 2-D kernels read the level rings or (opt-in since round 5) the cell-packed pair copies (include/parcels_hip.h: "velocity_pairs")
    fs._engine.ctx.set_option("velocity_pairs", 1 if pairs else 0)
    pset = build_pset(case, fs, sort_by_cell=sort)
    if case.get("populate", True):
        pset.populate_indices()
    kernels = [getattr(pa.kernels, k) for k in case["kernels"]]
    kw = {"endtime": endtime_of(endtime)} if endtime is not None else {"runtime": float(case["runtime"])}
    err = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            pset.execute(kernels, dt=float(case["dt"]), **kw)
        except (pa.FieldOutOfBoundError, pa.FieldOutOfBoundSurfaceError, pa.FieldInterpolationError, pa.GridSearchingError,
                pa.OutsideTimeInterval, pa.GeneralError) as e:
            err = type(e).__name__
    return {k: np.array(v) for k, v in pset._data.items()}, err, pset._last_stats


def _check(case, *, nslots=None, oracle=True, rtol=1e-12, endtime=None, probe=None, sort=False, expect_fast=True):
    fast, ferr, fstats = _run(case, True, nslots, endtime, probe, sort)
    gen, gerr, gstats = _run(case, False, nslots, endtime, probe, sort)
    assert ferr == gerr
    assert gstats["program"] != PROGRAM_FAST_CGRID
    assert (fstats["program"] == PROGRAM_FAST_CGRID) == expect_fast, fstats["program"]
    assert fstats["steps"] == gstats["steps"] and fstats["attempts"] == gstats["attempts"]
    # (float32 particle storage rounds every step's position to float32: a last-bit difference of the float64 sum can move the stored value by one float32 ulp)
    gtol = 5e-7 if case.get("spatial_dtype", "float64") == "float32" else FAST_VS_GENERAL_RTOL
    compare(fast, gen, rtol=gtol, atol_pos=gtol * _scale(case), check_state="all", label=case["name"] + ": fast vs general", skip=())
    if oracle:
        c = dict(case, populate=case.get("populate", True))
        ref, oerr, _ = run_oracle(c, endtime=endtime)
        assert ferr == oerr
        if ferr is None:
            # (like tests/test_gpu_fuzz.py and the bench-size checks: longitudes run through 0 on this mesh, so the yardstick is the
            # coordinate scale, |a - b| <= rtol * (|b| + scale))
            compare(fast, ref, rtol=rtol, atol_pos=rtol * _scale(case), check_state="all", label=case["name"] + ": fast vs oracle", skip=())
    return fast, fstats


@pytest.mark.parametrize("kernel", ["AdvectionRK4", "AdvectionRK4_3D"])
@pytest.mark.parametrize("fdt,sdt", [(np.float32, "float64"), (np.float64, "float64"), (np.float32, "float32"), (np.float64, "float32")])
def test_fast_equals_general_and_oracle(gpu, kernel, fdt, sdt):
    from oracle import cases

    case = cases.curv_cgrid_case("fastc_" + kernel, mesh="spherical", kernels=[kernel, "DeleteParticle"], seed=5, npart=3000, field_dtype=fdt,
                                 spatial_dtype=sdt, with_w=True, dt=3600.0, runtime=30 * 3600.0, vel=0.8)
    _check(case, rtol=5e-7 if sdt == "float32" else 1e-12)


@pytest.mark.parametrize("sdt", ["float64", "float32"])
@pytest.mark.parametrize("delete", [True, False])
def test_rk45_on_the_fast_evaluation_equals_the_general_program(gpu, sdt, delete):
    """AdvectionRK45 (adaptive dt, Repeat loop, next_dt column) through advect_cgrid_rk45_kernel: accepted and rejected attempts, dt
    halved and doubled between min_dt and max_dt, particles that leave the mesh with and without the recovery kernel (without it the
    launch is repeated with the iteration limit of the first error) -- the general program (discrete columns and counters exactly, positions to FAST_VS_GENERAL_RTOL) and the oracle to 1e-12."""
    from oracle import cases

    kernels = ["AdvectionRK45"] + (["DeleteParticle"] if delete else [])
    case = cases.curv_cgrid_case("fastc_rk45", mesh="spherical", kernels=kernels, seed=21, npart=2500, spatial_dtype=sdt, with_w=False, dt=1800.0,
                                 runtime=20 * 3600.0, vel=2.5 if delete else 0.4)
    case["context"] = {"RK45_tol": 30.0, "RK45_min_dt": 60.0, "RK45_max_dt": 4 * 3600.0}
    fast, st = _check(case, rtol=5e-7 if sdt == "float32" else 1e-12)
    assert st["attempts"] > st["steps"], "no attempt was rejected: the test does not test the Repeat loop"


@pytest.mark.parametrize("kernel,fdt,sdt", [("AdvectionRK4_3D", np.float32, "float64"), ("AdvectionRK45", np.float32, "float64"),
                                            ("AdvectionDiffusionM1", np.float32, "float64"), ("AdvectionRK4_3D", np.float64, "float32"),
                                            ("AdvectionRK4", np.float64, "float64"), ("AdvectionRK45", np.float64, "float32")])
def test_fine_mesh_takes_the_edge_cosines_from_the_sample_latitude(gpu, kernel, fdt, sdt):
    """A mesh whose cells all span less than 2^-8 rad of latitude (like BASELINE configs 3 - 5: 1/12 degree) runs the kernel variants
    that form the cosines of CGrid_Velocity's four edge latitudes from the sample point's own sine / cosine (FastC::near_edges,
    pk_fast_cgrid.h: cos_near); the coarse meshes of the other tests run the variants with the full cosine.  Oracle at 1e-12 (1e-11 with
    the Box-Muller draws), general program at FAST_VS_GENERAL_RTOL, cells crossed every other step."""
    from oracle import cases

    kw = dict(mesh="spherical", seed=41, nx=560, ny=420, nz=4, nt=3, npart=3000)
    if kernel == "AdvectionDiffusionM1":
        case = cases.curv_cgrid_diffusion_case("fastc_fine_m1", kernels=[kernel, "DeleteParticle"], dt=1800.0, **{k: v for k, v in kw.items() if k not in ("nz", "nt")})
    else:
        case = cases.curv_cgrid_case("fastc_fine_" + kernel, kernels=[kernel, "DeleteParticle"], with_w=kernel == "AdvectionRK4_3D", dt=3600.0,
                                     runtime=30 * 3600.0, vel=1.5, field_dtype=fdt, spatial_dtype=sdt, **kw)
        if kernel == "AdvectionRK45":
            case["context"] = {"RK45_tol": 30.0, "RK45_min_dt": 60.0, "RK45_max_dt": 4 * 3600.0}
    lat = np.asarray(case["lat"])
    ext = np.maximum.reduce([lat[:-1, :-1], lat[:-1, 1:], lat[1:, :-1], lat[1:, 1:]]) - np.minimum.reduce([lat[:-1, :-1], lat[:-1, 1:], lat[1:, :-1], lat[1:, 1:]])
    assert np.deg2rad(ext.max()) <= 2.0 ** -8, "the mesh is not fine enough to test the near-edges kernels"
    fast, st = _check(case, rtol=5e-7 if sdt == "float32" else (1e-11 if kernel == "AdvectionDiffusionM1" else 1e-12))
    assert st["steps"] > 10 * len(fast["x"])


def test_velocity_pairs_and_level_rings_give_the_same_bits(gpu):
    """The 2-D kernels read the staggered velocity from cell-packed pair copies (FastC::vp: both levels of a cell in one line) or, without
    the memory for them / with the option off, from the level rings: same values, same trajectories.  AdvectionRK45 on resident levels;
    AdvectionRK4 through a ring of 3 with release times spread over the levels, some exactly ON a level (the highest resident level has
    no pair of its own: upper half of the pair below), several launches."""
    from oracle import cases

    case = cases.curv_cgrid_case("fastc_pairs", mesh="spherical", kernels=["AdvectionRK45", "DeleteParticle"], seed=22, npart=2500, with_w=False,
                                 dt=1800.0, runtime=20 * 3600.0, vel=2.5)
    case["context"] = {"RK45_tol": 30.0, "RK45_min_dt": 60.0, "RK45_max_dt": 4 * 3600.0}
    on, err_on, st_on = _run(case, True, pairs=True)
    off, err_off, st_off = _run(case, True, pairs=False)
    assert err_on == err_off
    assert st_on["packs"] > 0 and st_on["pack_ms"] > 0 and st_off["packs"] == 0, (st_on, st_off)  # the copies were made (and timed) / were not
    assert st_on["program"] == PROGRAM_FAST_CGRID and st_off["program"] == PROGRAM_FAST_CGRID
    assert st_on["steps"] == st_off["steps"] and st_on["attempts"] == st_off["attempts"]
    compare(on, off, rtol=0.0, check_state="all", label="pair copies vs level rings", skip=())

    case = cases.curv_cgrid_case("fastc_pairs_ring", mesh="spherical", kernels=["AdvectionRK4", "DeleteParticle"], seed=23, nt=6, npart=4000, dt=3600.0,
                                 runtime=None, vel=1.5)
    n = len(case["x"])
    case["t0"] = np.random.default_rng(3).uniform(0, 3 * 86400.0, n)
    case["t0"][::5] = 86400.0 * (np.arange(len(case["t0"][::5])) % 3)  # some exactly on a level
    case["endtime"] = 4.5 * 86400.0
    case["runtime"] = None
    on, err_on, st_on = _run(case, True, nslots=3, endtime=case["endtime"], sort=True, pairs=True)
    off, err_off, st_off = _run(case, True, nslots=3, endtime=case["endtime"], sort=True, pairs=False)
    assert st_on["packs"] >= 2 and st_off["packs"] == 0
    assert err_on is None and err_off is None and st_on["launches"] > 1
    assert st_on["program"] == PROGRAM_FAST_CGRID and st_off["program"] == PROGRAM_FAST_CGRID
    compare(on, off, rtol=0.0, check_state="all", label="pair copies vs level rings, ring of 3", skip=())


@pytest.mark.parametrize("kh", ["node4d", "node2d"])
@pytest.mark.parametrize("sdt", ["float64", "float32"])
def test_m1_on_the_fast_evaluation_equals_the_general_program(gpu, kh, sdt):
    """AdvectionDiffusionM1 through advect_cgrid_m1_kernel: six scalar samples (Kh_zonal / Kh_meridional on the grid's nodes, 4-D with
    their own time / depth interpolation or 2-D) + one velocity sample per step, the `ei` guesses chained through all seven -- against
    the general program (FAST_VS_GENERAL_RTOL; it re-uses the velocity sample's grid position where the reference's renewed search would return it),
    the oracle to 1e-11 (Box-Muller's log / sin / cos)."""
    from oracle import cases

    case = cases.curv_cgrid_diffusion_case("fastc_m1", mesh="spherical", kernels=["AdvectionDiffusionM1", "DeleteParticle"], seed=31, npart=2500,
                                           spatial_dtype=sdt, kh=kh, dt=1800.0)
    fast, st = _check(case, rtol=5e-7 if sdt == "float32" else 1e-11)
    assert st["steps"] > 10 * len(fast["x"])


def test_cells_are_crossed_and_particles_leave_the_mesh(gpu):
    """Fast flow on a small mesh: most stages cross a cell edge (neighbour probe), many particles leave the mesh (the table walk
    finds nothing: GridSearchingError -> DeleteParticle) or the depth range."""
    from oracle import cases

    case = cases.curv_cgrid_case("fastc_fast_flow", mesh="spherical", kernels=["AdvectionRK4_3D", "DeleteParticle"], seed=9, nx=48, ny=36, nz=6, nt=4,
                                 npart=6000, dt=3600.0, runtime=60 * 3600.0, vel=6.0)
    fast, st = _check(case)
    assert len(fast["x"]) < 6000, "nothing left the mesh: the test does not test"


def test_errors_raise_the_same(gpu):
    from oracle import cases

    case = cases.curv_cgrid_case("fastc_raise", mesh="spherical", kernels=["AdvectionRK4_3D"], seed=10, nx=30, ny=24, nz=5, nt=3, npart=800, dt=3600.0,
                                 runtime=48 * 3600.0, vel=8.0)
    fast, ferr, fst = _run(case, True)
    gen, gerr, gst = _run(case, False)
    assert ferr == gerr and ferr is not None
    assert fst["program"] == PROGRAM_FAST_CGRID
    compare(fast, gen, rtol=FAST_VS_GENERAL_RTOL, atol_pos=FAST_VS_GENERAL_RTOL * _scale(case), check_state="all", label="raise", skip=())


@pytest.mark.parametrize("probe", [-1, 1])
def test_table_order_search_when_neighbour_probing_is_off(gpu, probe):
    """neighbour_probe = -1 (what a mesh with coincident nodes gets): every cell change goes through the hash-table walk."""
    from oracle import cases

    case = cases.curv_cgrid_case("fastc_walk", mesh="spherical", kernels=["AdvectionRK4_3D", "DeleteParticle"], seed=12, npart=2000, dt=3600.0,
                                 runtime=36 * 3600.0, vel=3.0)
    _check(case, probe=probe)


def test_level_ring_and_staggered_release_times(gpu):
    """Release times spread over the time levels (lanes of one wavefront on different levels: the waterfall of the field fetch
    iterates) through a ring of 3 levels (pause / resume, several launches), cell-sorted."""
    from oracle import cases

    case = cases.curv_cgrid_case("fastc_ring", mesh="spherical", kernels=["AdvectionRK4_3D", "DeleteParticle"], seed=13, nt=6, npart=4000, dt=3600.0,
                                 runtime=None, vel=1.5)
    n = len(case["x"])
    case["t0"] = np.random.default_rng(2).uniform(0, 3 * 86400.0, n)
    case["t0"][::5] = 86400.0 * (np.arange(len(case["t0"][::5])) % 3)  # some exactly on a level
    case["endtime"] = 4.5 * 86400.0
    case["runtime"] = None
    fast, fst = _check(case, endtime=case["endtime"])
    ring, rerr, rstats = _run(case, True, nslots=3, endtime=case["endtime"], sort=True)
    assert rerr is None and rstats["launches"] > 1 and rstats["program"] == PROGRAM_FAST_CGRID
    compare(ring, fast, rtol=0.0, check_state="all", label="ring", skip=())


def test_backward_in_time(gpu):
    from oracle import cases

    case = cases.curv_cgrid_case("fastc_back", mesh="spherical", kernels=["AdvectionRK4", "DeleteParticle"], seed=14, npart=2000, dt=-3600.0,
                                 runtime=40 * 3600.0, vel=2.0)
    case["t0"] = np.full(len(case["x"]), float(case["time_s"][-1]))
    _check(case)


def test_unguessed_first_launch_runs_the_general_program(gpu):
    """Without populate_indices() the first search of the reference has no guess and returns float32 (xsi, eta) arrays
    (spatialhash.py:505): that launch belongs to the general program; results equal with the option on or off."""
    from oracle import cases

    case = cases.curv_cgrid_case("fastc_unpop", mesh="spherical", kernels=["AdvectionRK4_3D", "DeleteParticle"], seed=15, npart=1500, dt=3600.0,
                                 runtime=12 * 3600.0, vel=1.0)
    case["populate"] = False
    _check(case, expect_fast=False, oracle=False)


def test_flat_mesh_is_not_eligible(gpu):
    from oracle import cases

    case = cases.curv_cgrid_case("fastc_flat", mesh="flat", kernels=["AdvectionRK4_3D", "DeleteParticle"], seed=16, npart=500, dt=3600.0,
                                 runtime=6 * 3600.0, vel=1.0)
    _check(case, expect_fast=False, oracle=False)
