"""Helpers shared by the parity tests: turn a neutral case dict (oracle/cases.py, tests/golden/*.npz) into
parcels_amd objects, run it through the HIP path, and compare particle SoA dicts."""

from __future__ import annotations

import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def golden_names(kind="advect"):
    """Fixture names: "advect" = ParticleSet.execute trajectories (all of them), "sample" = Field.eval at explicit points,
    "v3jit" = the subset of "advect" built from the reference's v3-JIT regression data (oracle/make_v3_golden.py)."""
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    if kind == "v3jit":
        return [n for n in names if n.startswith("v3jit_")]
    return [n for n in names if n.startswith("sample_") == (kind == "sample")]


def load_golden(name):
    from oracle import make_golden as mg

    return mg.load_case(os.path.join(GOLDEN_DIR, name + ".npz"))


def is_curvilinear(case):
    return np.asarray(case["lon"]).ndim == 2


def attach_hash_table(case):
    """Build the spatial hash with the framework's own host builder and pin it to the reference's checksum."""
    from parcels_amd import spatialhash as sh

    h = sh.SpatialHash(case["lon"], case["lat"], case["mesh"] == "spherical")
    if "hash_checksum" in case:
        assert h.checksum() == case["hash_checksum"], "spatial-hash table differs from the reference's"
    case["hash_table"] = h.table()
    return case


def stop_time_of_reference(case, out, err):
    """The batch reference stops at the iteration of the first error; per-particle engines are compared there."""
    if err is None or err == "OutsideTimeInterval" or len(out["t"]) == 0:
        return None
    return float(np.max(out["t"]) if case["dt"] > 0 else np.min(out["t"]))


def build_fieldset(case):
    import parcels_amd as pa

    pad = {"low": pa.Padding.LOW, "high": pa.Padding.HIGH, "both": pa.Padding.BOTH, "none": pa.Padding.NONE}
    depth = case.get("depth")
    md = pa.SGrid2DMetadata(
        node_dimensions=("XG", "YG"),
        node_coordinates=("lon", "lat"),
        face_dimensions=(
            pa.FaceNodePadding("XC", "XG", pad[case.get("x_pad", "low")]),
            pa.FaceNodePadding("YC", "YG", pad[case.get("y_pad", "low")]),
        ),
        vertical_dimensions=(pa.FaceNodePadding("ZC", "depth", pad[case.get("z_pad", "both")]),) if depth is not None else None,
    )
    lon, lat = np.asarray(case["lon"]), np.asarray(case["lat"])
    coords = {}
    if lon.ndim == 1:
        coords["lon"] = (("XG",), lon)
        coords["lat"] = (("YG",), lat)
    else:
        coords["lon"] = (("YG", "XG"), lon)
        coords["lat"] = (("YG", "XG"), lat)
    if depth is not None:
        coords["depth"] = (("depth",), np.asarray(depth))
    ts = case.get("time_s")
    if ts is not None and len(ts) > 1:
        coords["time"] = (("time",), np.asarray(ts, dtype=np.float64))
    data_vars = {}
    for name, arr in case["fields"].items():
        dims = tuple(case["field_dims"][name])
        if hasattr(arr, "read_level"):  # a level source (parcels_amd.sources): handed over as it is
            data_vars[name] = (dims, arr)
            continue
        a = np.asarray(arr)
        keep = [i for i, d in enumerate(dims) if not d.startswith("mock")]
        a = a.reshape([a.shape[i] for i in keep])
        data_vars[name] = (tuple(dims[i] for i in keep), a)
    ds = pa.Dataset(data_vars, coords, sgrid=md)
    fs = pa.FieldSet.from_sgrid_conventions(ds, mesh=case["mesh"], vector_fields={} if case.get("kind") == "sample" else None)
    if case.get("cgrid"):  # node-dimensioned fields read as a C-grid (the reference's v3 regression test, test_interpolation.py:346-347)
        for vname in ("UV", "UVW"):
            if vname in fs.fields and not isinstance(fs.fields[vname].interp_method, pa.CGrid_Velocity):
                fs.fields[vname].interp_method = pa.CGrid_Velocity()
    if case.get("slip"):
        interp = {"free": pa.XFreeslip, "partial": pa.XPartialslip}[case["slip"]]()
        for vname in ("UV", "UVW"):
            if vname in fs.fields:
                fs.fields[vname].interp_method = interp
    for name, val in (case.get("constants") or {}).items():
        fs.add_constant_field(name, val, mesh=case.get("const_mesh", "flat"))
    for k, v in (case.get("context") or {}).items():
        fs.add_context(k, v)
    return fs


def sample_hip(case):
    """Field.eval through pk_eval with the requested scalar interpolator."""
    import parcels_amd as pa

    fs = build_fieldset(case)
    name = case["sample_field"]
    fs.fields[name].interp_method = getattr(pa, case["scalar_interp"][name])()
    fs._engine = None  # the interpolator is part of the device field descriptor
    val = fs.fields[name].eval(case["t0"], case["z"], case["y"], case["x"])
    return {"value": np.asarray(val)}


def build_pset(case, fs, **kw):
    import parcels_amd as pa

    sdt = np.dtype(case.get("spatial_dtype", "float64")).type
    pclass = pa.get_default_particle(sdt)
    if "AdvectionRK45" in case["kernels"]:
        pclass = pclass.add_variable(pa.Variable("next_dt", dtype=np.dtype(case.get("next_dt_dtype", "float64")).type,
                                                 initial=float(case.get("next_dt0", case["dt"]))))
    for fname, vname, vdt in (case.get("sample_into") or {}).values():
        for vn in (vname if isinstance(vname, (list, tuple)) else [vname]):
            if vn is not None:
                pclass = pclass.add_variable(pa.Variable(vn, dtype=np.dtype(vdt).type, initial=0))
    n = len(np.atleast_1d(case["x"]))
    t0 = case.get("t0")
    t = np.zeros(n) if t0 is None else np.broadcast_to(np.asarray(t0, dtype=np.float64), (n,)).copy()
    return pa.ParticleSet(fs, pclass=pclass, x=np.asarray(case["x"]), y=np.asarray(case["y"]), z=case.get("z"), t=t,
                          seed=int(case.get("seed", 0)), **kw)


def golden_observation(out, k):
    """Observation k of a fixture: (ids, t, z, y, x) -- rectangular arrays, or ragged ones (deletions between the output times) stored back
    to back with `obs_offsets` (oracle/ref_shim.py)."""
    if "obs_offsets" in out:
        sl = slice(int(out["obs_offsets"][k]), int(out["obs_offsets"][k + 1]))
        return tuple(out["obs_" + c][sl] for c in ("particle_id", "t", "z", "y", "x"))
    return tuple(out["obs_" + c][k] for c in ("particle_id", "t", "z", "y", "x"))


class OutputRecorder:
    """Duck-typed ParticleFile: makes ParticleSet.execute split the run into output intervals (particleset.py:419-462) and
    records the observations in memory."""

    def __init__(self, outputdt):
        self.outputdt = outputdt
        self.metadata = {}
        self.obs = []

    def set_metadata(self, mesh):
        pass

    def write(self, pset, time):
        self.obs.append((float(time), np.array(pset._data["particle_id"]), np.array(pset._data["x"]), np.array(pset._data["y"]),
                         np.array(pset._data["z"]), np.array(pset._data["t"])))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def endtime_of(seconds):
    """Seconds since the start of a float-second time axis as the `endtime` ParticleSet.execute accepts: the type of
    fieldset.time_interval.left (particleset.py:553-557), the way oracle/ref_shim.py hands it to the reference."""
    return np.timedelta64(int(round(float(seconds) * 1e9)), "ns")


def run_hip(case, endtime=None, nslots=None, async_output=None, fieldset=None, **pset_kw):
    """Run a case through parcels_amd (HIP). Returns (soa dict, error name or None, stats).  nslots: stream the field levels
    through a ring of that many slots (None: the engine decides, i.e. resident for test-sized fields)."""
    import parcels_amd as pa

    fs = fieldset if fieldset is not None else build_fieldset(case)
    if nslots is not None:
        fs.to_device(nslots=nslots)
    pset = build_pset(case, fs, **pset_kw)
    if async_output is not None:
        pset.async_output = bool(async_output)
    samples = case.get("sample_into") or {}
    kernels = [pa.SampleField(samples[k][0], into=tuple(samples[k][1]) if isinstance(samples[k][1], (list, tuple)) else samples[k][1])
               if k in samples else getattr(pa.kernels, k) for k in case["kernels"]]
    kw = {}
    if endtime is not None:
        kw["endtime"] = endtime_of(endtime)
    elif case.get("endtime") is not None:
        kw["endtime"] = endtime_of(case["endtime"])
    else:
        kw["runtime"] = float(case["runtime"])
    run_hip.last_recorder = None
    if case.get("outputdt"):
        kw["output_file"] = run_hip.last_recorder = OutputRecorder(float(case["outputdt"]))
    err = None
    import warnings

    if case.get("populate"):
        pset.populate_indices()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            pset.execute(kernels, dt=float(case["dt"]), **kw)
            for call in case.get("more_calls") or ():  # further execute() calls on the same set (device-resident columns in between)
                pset.execute(kernels, dt=float(call["dt"]), runtime=float(call["runtime"]))
        except (pa.FieldOutOfBoundError, pa.FieldOutOfBoundSurfaceError, pa.FieldInterpolationError, pa.GridSearchingError,
                pa.OutsideTimeInterval, pa.GeneralError) as e:
            err = type(e).__name__
    return {k: np.array(v) for k, v in pset._data.items()}, err, pset._last_stats


def run_oracle(case, endtime=None, nthreads=1, call_wide_time_error=True):
    from oracle import c_oracle as co

    c = dict(case)
    if endtime is not None:
        c["endtime"] = float(endtime_of(endtime) / np.timedelta64(1, "s"))  # what run_hip's execute derives from the same argument
        c["runtime"] = None
    if is_curvilinear(c) and "hash_table" not in c:
        attach_hash_table(c)
    return co.run_case(c, nthreads=nthreads, call_wide_time_error=call_wide_time_error)


def max_rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    d = np.abs(a - b)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = np.where(d == 0, 0.0, d / np.maximum(np.abs(b), 1e-300))
    return float(np.nanmax(r))


def compare(got, ref, *, rtol, atol_pos=0.0, check_state="all", err_mask=None, skip=("dt",), label=""):
    """Compare two particle SoA dicts. Positions to rtol (relative to |ref|) or atol_pos; t, ei, ids exactly."""
    assert len(got["x"]) == len(ref["x"]), f"{label}: particle count {len(got['x'])} != {len(ref['x'])}"
    assert np.array_equal(got["particle_id"], ref["particle_id"]), f"{label}: particle order differs"
    report = {}
    user = [k for k in ref if k not in ("x", "y", "z", "dx", "dy", "dz", "next_dt", "dt", "t", "state", "ei", "particle_id") and not k.startswith("obs_")]
    for k in ["x", "y", "z", "dx", "dy", "dz", "next_dt", "dt"] + user:
        if k in skip or k not in ref or k not in got:
            continue
        a, b = np.asarray(got[k], dtype=np.float64), np.asarray(ref[k], dtype=np.float64)
        ok = np.isclose(a, b, rtol=rtol, atol=atol_pos, equal_nan=True)
        report[k] = max_rel(a, b)
        assert ok.all(), f"{label}: {k} differs: max rel {report[k]:.3e} (rtol {rtol:g}) at {np.flatnonzero(~ok)[:5]}"
    assert np.array_equal(got["t"], ref["t"]), f"{label}: t differs"
    if check_state == "all":
        assert np.array_equal(got["state"], ref["state"]), f"{label}: state differs: {np.flatnonzero(got['state'] != ref['state'])[:8]}"
    elif check_state == "errors":
        m = ref["state"] >= 50 if err_mask is None else err_mask
        assert np.array_equal(got["state"][m], ref["state"][m]), f"{label}: error states differ"
    assert np.array_equal(got["ei"], ref["ei"]), f"{label}: ei differs at {np.flatnonzero((got['ei'] != ref['ei']).any(axis=1))[:8]}"
    return report


# tolerance class of every golden case (see DESIGN.md "Parity"); the north star's bar is 1e-6 relative
# Fourth tolerance class, for runs whose FIRST curvilinear search has no guess (no populate_indices()): the reference then takes
# (xsi, eta) from the float32 buffer of the hash query (spatialhash.py:505).  The device's sin / cos of the query point differ from
# the oracle's libm by <= 1 ulp of float64 (both are <= 1 ulp routines; neither is the reference's NumPy SIMD loop either), and where
# that ulp straddles a float32 rounding boundary of xsi or eta the rounded value moves by one float32 ulp (6e-8): a handful of
# particles out of 50 000 start 6e-8 of a cell width (~1e-9 of the coordinate) apart and keep that offset.  It is a discontinuity of
# the reference itself (float32 rounding of a float64 intermediate), not an accumulation: it cannot be removed without bit-identical
# transcendentals, so it is bounded and named here.  Populated runs (every search guessed) are in the 1e-12 class.
UNGUESSED_CURVILINEAR_RTOL = 1e-7


def count_outside(got, ref, rtol, scale=0.0):
    """Particles (rows present in both, same order) with a position component further than rtol * (|ref| + scale) from the reference."""
    bad = np.zeros(len(ref["x"]), bool)
    for k in ("x", "y", "z"):
        a, b = np.asarray(got[k], np.float64), np.asarray(ref[k], np.float64)
        with np.errstate(invalid="ignore"):
            bad |= np.abs(a - b) > rtol * (np.abs(b) + scale)
    return int(bad.sum())


def tolerance_for(name, case):
    if case.get("spatial_dtype", "float64") == "float32":
        # one float32 ulp of the stored position: the float32 cos() of the first stage (device cosf vs NumPy's) may differ by an
        # ulp, which now and then flips the float32 rounding of a stored coordinate.  NumPy's float32 dtype propagation itself
        # (float32 coordinates / barycentric arrays / data) is reproduced operation by operation (pk_device.h: TYPED, NV pairs).
        return 5e-7
    if any(k.startswith("AdvectionDiffusion") or k == "DiffusionUniformKh" for k in case["kernels"]):
        return 1e-11  # log/sin/cos of the Box-Muller transform differ by an ulp between libm implementations
    return 1e-12


# ---- the reference's v3-JIT regression trajectories (tests/test_interpolation.py:297-378) ---------------------------
V3_ATOL = 1e-6  # the reference's own bar for v4-vs-v3 (np.testing.assert_allclose(..., atol=1e-6), :376-378)


def v3_observations(run, case):
    """lon/lat/z (n, 4) at t = 0, 1, 2, 3 s from `run(case, endtime) -> soa dict`; deleted particles are NaN like in the v3 file."""
    n = len(case["x"])
    sdt = np.dtype(case["spatial_dtype"])
    obs = {k: np.full((n, 4), np.nan) for k in ("x", "y", "z")}
    for k in obs:
        obs[k][:, 0] = np.asarray(case[k]).astype(sdt)
    for step in (1, 2, 3):
        out = run(case, float(step))
        ids = np.asarray(out["particle_id"])
        for k in obs:
            obs[k][ids, step] = out[k]
    return obs


def assert_matches_v3(obs, case, label):
    for k, ref in (("x", "v3_lon"), ("y", "v3_lat"), ("z", "v3_z")):
        v3 = np.asarray(case[ref])
        assert np.array_equal(np.isnan(obs[k]), np.isnan(v3)), f"{label}: the set of deleted particles differs from v3 ({k})"
        np.testing.assert_allclose(obs[k], v3, atol=V3_ATOL, rtol=0, equal_nan=True, err_msg=f"{label}: {k}")
