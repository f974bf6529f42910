"""GPU: seeded differential fuzzing of the HIP path against the CPU oracle.

Every seed draws a grid type (rectilinear A-grid / rectilinear C-grid / curvilinear A- or C-grid), mesh, field / particle /
coordinate dtypes, a kernel list (Euler .. RK45, Milstein / Euler-Maruyama diffusion, recovery kernels), time direction,
staggered releases, release margins that push particles out of the domain, output intervals and the device cell sort, and
demands what the fixed cases demand: same raised error, same `t`, `state`, `ei` and particle order exactly, positions to the
tolerance class of the configuration (tests/case_utils.py).  The oracle itself is pinned to the reference bit for bit
(tests/test_oracle_golden.py), so this widens the reference-parity net to configurations no fixture holds.
"""

import os

import numpy as np
import pytest

from case_utils import compare, run_hip, run_oracle

pytestmark = pytest.mark.gpu

ADVECTION_2D = ["AdvectionEE", "AdvectionRK2", "AdvectionRK4", "AdvectionRK45"]
ADVECTION_3D = ["AdvectionRK2_3D", "AdvectionRK4_3D"]
RECOVERY = [None, "DeleteParticle", "DeleteOutOfBounds", "SubmergeParticle"]


def draw_case(seed):
    from oracle import cases

    rng = np.random.default_rng(1000 + seed)
    kind = rng.choice(["agrid", "agrid", "cgrid", "curv_c", "curv_a", "diffusion", "curv_diffusion", "slip"])
    mesh = str(rng.choice(["spherical", "flat"]))
    backward = bool(rng.random() < 0.25)
    npart = int(rng.integers(200, 1500))
    three_d = bool(rng.random() < 0.5)
    kernel = str(rng.choice(ADVECTION_3D if three_d else ADVECTION_2D))
    kernels = [kernel]
    if rng.random() < 0.3 and kernel != "AdvectionRK45":
        kernels.append(str(rng.choice(ADVECTION_2D[:3] + (ADVECTION_3D if three_d else []))))
    rec = RECOVERY[int(rng.integers(0, len(RECOVERY)))]
    if rec == "SubmergeParticle":
        kernels += ["SubmergeParticle", "DeleteOutOfBounds"]
    elif rec:
        kernels.append(rec)
    sdt = str(rng.choice(["float64", "float64", "float32"]))
    fdt = np.float32 if rng.random() < 0.5 else np.float64
    dt = float(rng.choice([600.0, 1800.0, 3600.0])) * (-1 if backward else 1)
    nsteps = int(rng.integers(5, 30))
    common = dict(mesh=mesh, kernels=kernels, seed=int(seed), npart=npart)
    if kind == "agrid":
        cdt = np.float32 if (rng.random() < 0.25) else np.float64
        case = cases.rect_agrid_case(f"fuzz{seed}", **common, nx=int(rng.integers(8, 40)), ny=int(rng.integers(6, 30)),
                                     nz=int(rng.integers(2, 8)), nt=int(rng.integers(2, 5)), field_dtype=fdt, spatial_dtype=sdt,
                                     coord_dtype=cdt, with_w=three_d, dt=dt, runtime=nsteps * abs(dt),
                                     margin=float(rng.choice([0.15, 0.05, 0.01])), vel=float(rng.choice([0.5, 2.0, 6.0])),
                                     stagger=bool(rng.random() < 0.3))
    elif kind == "cgrid":
        case = cases.rect_cgrid_case(f"fuzz{seed}", **common, field_dtype=fdt, spatial_dtype=sdt)
        case["dt"], case["runtime"] = dt, nsteps * abs(dt)
    elif kind in ("curv_c", "curv_a"):
        case = cases.curv_cgrid_case(f"fuzz{seed}", **common, nx=int(rng.integers(20, 60)), ny=int(rng.integers(16, 45)),
                                     field_dtype=fdt, spatial_dtype=sdt, with_w=True, dt=dt, runtime=nsteps * abs(dt),
                                     vel=float(rng.choice([0.3, 1.0, 3.0])), cgrid=(kind == "curv_c"))
        case["populate"] = bool(rng.random() < 0.5)
    elif kind == "curv_diffusion":  # BASELINE config 5: Milstein / Euler-Maruyama on the 3-D curvilinear C-grid, Kh on its nodes
        dk = str(rng.choice(["AdvectionDiffusionM1", "AdvectionDiffusionM1", "AdvectionDiffusionEM"]))
        case = cases.curv_cgrid_diffusion_case(f"fuzz{seed}", mesh=mesh, kernels=[dk] + ([rec] if rec in ("DeleteParticle", "DeleteOutOfBounds") else []),
                                               seed=int(seed), npart=npart, spatial_dtype=sdt, kh=str(rng.choice(["node4d", "2d"])),
                                               nx=int(rng.integers(20, 60)), ny=int(rng.integers(16, 45)), field_dtype=fdt,
                                               kh_dtype=np.float32 if rng.random() < 0.5 else np.float64, dt=abs(dt), runtime=nsteps * abs(dt))
        case["populate"] = bool(rng.random() < 0.5)
        backward = False
    elif kind == "slip":  # XFreeslip / XPartialslip around land blocks
        k2 = [str(rng.choice(["AdvectionRK4_3D", "AdvectionRK2_3D"] if three_d else ["AdvectionEE", "AdvectionRK2", "AdvectionRK4"]))]
        case = cases.slip_case(f"fuzz{seed}", slip=str(rng.choice(["free", "partial"])), mesh=mesh, kernels=k2 + (["DeleteParticle"] if three_d else []),
                               seed=int(seed), with_w=three_d, npart=npart, spatial_dtype=sdt, field_dtype=fdt)
        case["dt"], case["runtime"] = dt, nsteps * abs(dt)
    else:
        dk = str(rng.choice(["AdvectionDiffusionM1", "AdvectionDiffusionEM"]))
        case = cases.diffusion_case(f"fuzz{seed}", mesh=mesh, kernels=[dk] + ([rec] if rec in ("DeleteParticle", "DeleteOutOfBounds") else []),
                                    seed=int(seed), npart=npart, spatial_dtype=sdt, const_kh=(5.0 if rng.random() < 0.3 else None))
        case["dt"] = abs(case["dt"])
        backward = False
    if "AdvectionRK45" in case["kernels"]:
        case["context"] = {"RK45_tol": float(rng.choice([0.5, 50.0, 5000.0])), "RK45_min_dt": 10.0, "RK45_max_dt": 4 * 3600.0}
        if rng.random() < 0.5:
            case["next_dt_dtype"] = "float32"
    if backward and case.get("time_s") is not None:
        case["t0"] = np.full(len(case["x"]), float(case["time_s"][-1]))
        case["dt"] = -abs(case["dt"])
    if rng.random() < 0.3 and kind not in ("diffusion", "curv_diffusion"):
        case["outputdt"] = float(abs(case["dt"]) * rng.choice([2.0, 2.5, 3.7]))
    return case, bool(rng.random() < 0.5)


def tolerance(case):
    if case.get("spatial_dtype", "float64") == "float32":
        return 5e-7  # one float32 ulp of the stored position
    if np.asarray(case["lon"]).ndim == 2 and not case.get("populate"):
        return 1e-10  # unguessed first evaluation: (xsi, eta) rounded to float32 like the reference's (spatialhash.py:505)
    return 1e-11


def draw_engine_options(seed, case):
    """How the engine runs the case (a second, independent stream: the cases of draw_case stay what they were): field levels
    resident or streamed through a ring of 3 / 4 slots.  None of it may change a result."""
    rng = np.random.default_rng(77000 + seed)
    ts = case.get("time_s")
    nt = len(ts) if ts is not None else 1
    nslots = None
    if nt >= 4 and rng.random() < 0.8:
        nslots = int(rng.choice([3, 4])) if nt > 4 else 3
    return dict(nslots=nslots)


def decorate_case(seed, case):
    """User-kernel tokens on top of the drawn configuration (a third independent stream): with probability 0.3 a vector sample
    `particles.u, particles.v[, particles.w] = fieldset.UV[W][particles]` (float32 or float64 Variables, sometimes a discarded
    component) is inserted before or after the advection kernel; now and then MoveEast / DoNothing join the list."""
    rng = np.random.default_rng(99000 + seed)
    if rng.random() < 0.3 and "AdvectionRK45" not in case["kernels"]:
        three_d = "W" in case["fields"] and rng.random() < 0.5
        names = ["u", "v", "w"][: 3 if three_d else 2]
        if rng.random() < 0.3:
            names[int(rng.integers(0, len(names)))] = None
        if all(n is None for n in names):
            names[0] = "u"
        token = "SampleUVW" if three_d else "SampleUV"
        case["kernels"].insert(int(rng.integers(0, 2)), token)
        case["sample_into"] = {token: ["UVW" if three_d else "UV", names, str(rng.choice(["float32", "float64"]))]}
    if rng.random() < 0.15:
        case["kernels"].insert(0, str(rng.choice(["MoveEast", "DoNothing", "MoveNorth"])))
    return case


_SEED0 = int(os.environ.get("PARCELS_FUZZ_SEED0", "0"))  # sweeps beyond the default range: PARCELS_FUZZ_SEED0=12000 PARCELS_FUZZ_SEEDS=20000


@pytest.mark.parametrize("seed", range(_SEED0, _SEED0 + int(os.environ.get("PARCELS_FUZZ_SEEDS", "256"))))
def test_random_configuration_matches_oracle(gpu, seed):
    case, sort_by_cell = draw_case(seed)
    case = decorate_case(seed, case)
    opts = draw_engine_options(seed, case)
    ref, oerr, ostats = run_oracle(case)
    got, gerr, stats = run_hip(case, sort_by_cell=sort_by_cell, **opts)
    label = f"seed {seed}: {case['kernels']} mesh={case['mesh']} lon{np.asarray(case['lon']).shape} dt={case['dt']} sort={sort_by_cell} " \
            f"outputdt={case.get('outputdt')} sdt={case.get('spatial_dtype')} {opts}"
    assert gerr == oerr, label
    # positions near 0 (a longitude crossing the Greenwich meridian) carry the absolute rounding noise of the coordinate scale
    tol = tolerance(case)
    scale = float(max(np.nanmax(np.abs(case["lon"])), np.nanmax(np.abs(case["lat"]))))
    compare(got, ref, rtol=tol, atol_pos=tol * scale, check_state="all", label=label)


SAMPLERS = [("XLinear", {}), ("XNearest", {}), ("CGrid_Tracer", {"zpad": "high"}), ("XLinearInvdistLandTracer", {"uniform_batch": "interior"}),
            ("XLinearInvdistLandTracer", {"uniform_batch": "t0"})]


@pytest.mark.parametrize("seed", range(int(os.environ.get("PARCELS_FUZZ_SAMPLE_SEEDS", "30"))))
def test_random_sampling_matches_oracle(gpu, seed):
    """Field.eval (pk_eval) with every scalar interpolator on random points (interior, exact nodes, land blocks, outside the
    domain), f32 / f64 data, flat / spherical: equal to the oracle, which an offline sweep of 400 such cases found bit-identical
    to the reference."""
    from case_utils import sample_hip
    from oracle import c_oracle as co
    from oracle import cases

    interp, kw = SAMPLERS[seed % len(SAMPLERS)]
    case = cases.sample_case(f"fuzz_sample{seed}", interp=interp, mesh="spherical" if (seed // 5) % 2 else "flat", seed=500 + seed, npts=500,
                             field_dtype=np.float32 if (seed // 10) % 2 else np.float64, **kw)
    got = np.asarray(sample_hip(case)["value"])
    ref = np.asarray(co.sample_case(case)["value"])
    np.testing.assert_allclose(got, ref, rtol=1e-13, atol=0, equal_nan=True)
