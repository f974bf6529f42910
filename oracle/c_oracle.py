"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of ``oracle/parcels_oracle.c`` (the CPU oracle).

A *case* is a plain dict (see ``oracle/cases.py``) describing grid, fields, particles and the execute() call in
NumPy terms.  ``run_case`` marshals it into the C structs and runs ``po_execute``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
PO_MAX_TWE = 1024  # po_params.twe_key (parcels_oracle.c)

KERNEL_IDS = {
    "AdvectionEE": 1,
    "AdvectionRK2": 2,
    "AdvectionRK2_3D": 3,
    "AdvectionRK4": 4,
    "AdvectionRK4_3D": 5,
    "AdvectionRK45": 6,
    "AdvectionDiffusionM1": 7,
    "AdvectionDiffusionEM": 8,
    "DiffusionUniformKh": 9,
    "SampleField": 10,
    "DoNothing": 23,
    "MoveEast": 24,
    "MoveNorth": 25,
    "DeleteParticle": 20,
    "DeleteOutOfBounds": 21,
    "SubmergeParticle": 22,
}

SCALAR_INTERP = {"XLinear": 0, "XConstantField": 1, "XNearest": 2, "CGrid_Tracer": 3, "XLinearInvdistLandTracer": 4}

EARTH_RADIUS = 6366707.019493707  # mesh.py:6


class PoGrid(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("spherical", C.c_int32),
        ("has_x", C.c_int32),
        ("has_y", C.c_int32),
        ("has_z", C.c_int32),
        ("nx", C.c_int32),
        ("ny", C.c_int32),
        ("nz", C.c_int32),
        ("xdim", C.c_int32),
        ("ydim", C.c_int32),
        ("zdim", C.c_int32),
        ("off_x", C.c_int32),
        ("off_y", C.c_int32),
        ("off_z", C.c_int32),
        ("lon_f32", C.c_int32),
        ("lat_f32", C.c_int32),
        ("depth_f32", C.c_int32),
        ("pad0", C.c_int32),
        ("deg2m", C.c_double),
        ("lon", C.c_void_p),
        ("lat", C.c_void_p),
        ("depth", C.c_void_p),
        ("dlon", C.c_void_p),
        ("dlat", C.c_void_p),
        ("ddepth", C.c_void_p),
        ("h_keys", C.c_void_p),
        ("h_starts", C.c_void_p),
        ("h_counts", C.c_void_p),
        ("h_faces", C.c_void_p),
        ("h_nkeys", C.c_int64),
        ("h_bitwidth", C.c_int32),
        ("pad1", C.c_int32),
        ("h_bbox", C.c_double * 6),
    ]


class PoField(C.Structure):
    _fields_ = [
        ("grid", C.c_int32),
        ("dtype", C.c_int32),
        ("nt", C.c_int32),
        ("nz", C.c_int32),
        ("ny", C.c_int32),
        ("nx", C.c_int32),
        ("has_t", C.c_int32),
        ("has_z", C.c_int32),
        ("has_y", C.c_int32),
        ("has_x", C.c_int32),
        ("has_time_interval", C.c_int32),
        ("is_const", C.c_int32),
        ("data", C.c_void_p),
        ("time", C.c_void_p),
    ]


class PoParticles(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("ngrids", C.c_int32),
        ("spatial_f32", C.c_int32),
        ("t", C.c_void_p),
        ("z", C.c_void_p),
        ("y", C.c_void_p),
        ("x", C.c_void_p),
        ("dz", C.c_void_p),
        ("dy", C.c_void_p),
        ("dx", C.c_void_p),
        ("dt", C.c_void_p),
        ("next_dt", C.c_void_p),
        ("state", C.c_void_p),
        ("ei", C.c_void_p),
        ("particle_id", C.c_void_p),
        ("extra", C.c_void_p * 4),
        ("extra_f32", C.c_int32 * 4),
    ]


class PoParams(C.Structure):
    _fields_ = [
        ("nk", C.c_int32),
        ("kernels", C.c_int32 * 8),
        ("cgrid", C.c_int32),
        ("rk45_mode", C.c_int32),
        ("have_guess0", C.c_int32),
        ("fU", C.c_int32),
        ("fV", C.c_int32),
        ("fW", C.c_int32),
        ("fKhz", C.c_int32),
        ("fKhm", C.c_int32),
        ("next_dt_f32", C.c_int32),
        ("force_lent", C.c_int32),
        ("force_lenz", C.c_int32),
        ("max_iters", C.c_int32),
        ("sample_field", C.c_int32 * 8),
        ("sample_var", C.c_int32 * 8),
        ("endtime", C.c_double),
        ("dt0", C.c_double),
        ("rk45_tol", C.c_double),
        ("rk45_min_dt", C.c_double),
        ("rk45_max_dt", C.c_double),
        ("dres", C.c_double),
        ("seed", C.c_uint64),
        ("twe_key", C.c_int64 * PO_MAX_TWE),
    ]


class PoStats(C.Structure):
    _fields_ = [("steps", C.c_int64), ("attempts", C.c_int64), ("first_error_iter", C.c_int64), ("first_time_error_key", C.c_int64)]


def build(force: bool = False) -> str:
    """Compile oracle/liboracle.so with gcc (building the checker is not using it)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("parcels_oracle.c", "fast_agrid_cpu.c")]
    if force or not os.path.exists(so) or any(os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so) for src in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.po_execute.restype = C.c_int
        _LIB.po_sizeof.restype = C.c_int
        assert _LIB.po_sizeof(0) == C.sizeof(PoGrid), (_LIB.po_sizeof(0), C.sizeof(PoGrid))
        assert _LIB.po_sizeof(1) == C.sizeof(PoField)
        assert _LIB.po_sizeof(2) == C.sizeof(PoParticles)
        assert _LIB.po_sizeof(3) == C.sizeof(PoParams)
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def fast_rk4_agrid(case: dict, endtime: float, nthreads: int = 1, sort_by_cell: bool = True):
    """AdvectionRK4 of `case` (rectilinear A-grid, float64 everything, particles released at t = 0) to `endtime` through
    oracle/fast_agrid_cpu.c, the CPU-idiomatic restatement of the headline workload (hinted searches, no dtype emulation, OpenMP).
    Returns ({"t", "y", "x", "state"} in the caller's particle order, steps, seconds of the timed call).  sort_by_cell: the particles
    are handed over ordered by cell (what the GPU path does on the device, outside ITS timed region too)."""
    import time as _time

    lon, lat, depth = (np.ascontiguousarray(case[k], dtype=np.float64) for k in ("lon", "lat", "depth"))
    tm = np.ascontiguousarray(case["time_s"], dtype=np.float64)
    U = np.ascontiguousarray(case["fields"]["U"], dtype=np.float64)
    V = np.ascontiguousarray(case["fields"]["V"], dtype=np.float64)
    assert U.shape == (len(tm), len(depth), len(lat), len(lon)) == V.shape
    x, y, z = (np.array(case[k], dtype=np.float64) for k in ("x", "y", "z"))
    n = len(x)
    order = np.arange(n)
    if sort_by_cell:
        zi = np.clip(np.searchsorted(depth, z) - 1, 0, max(len(depth) - 2, 0))
        yi = np.clip(np.searchsorted(lat, y) - 1, 0, max(len(lat) - 2, 0))
        xi = np.clip(np.searchsorted(lon, x) - 1, 0, max(len(lon) - 2, 0))
        order = np.argsort((zi * len(lat) + yi) * len(lon) + xi, kind="stable")
        x, y, z = x[order], y[order], z[order]
    t = np.zeros(n)
    state = np.zeros(n, np.int32)
    steps = C.c_int64(0)
    fn = lib().pf_rk4_agrid
    fn.restype = C.c_int64
    t0 = _time.perf_counter()
    bad = fn(_ptr(lon), C.c_int(len(lon)), _ptr(lat), C.c_int(len(lat)), _ptr(depth), C.c_int(len(depth)), _ptr(tm), C.c_int(len(tm)), _ptr(U), _ptr(V),
             C.c_int(int(case["mesh"] == "spherical")), C.c_double(EARTH_RADIUS * np.pi / 180.0), C.c_int64(n), _ptr(t), _ptr(z), _ptr(y), _ptr(x), _ptr(state),
             C.c_double(float(case["dt"])), C.c_double(float(endtime)), C.c_int(int(nthreads)), C.byref(steps))
    el = _time.perf_counter() - t0
    if bad:
        raise ValueError(f"{bad} particles leave the domain / time interval: fast_agrid_cpu.c covers the in-bounds workload only")
    inv = np.empty(n, np.int64)
    inv[order] = np.arange(n)
    return {"t": t[inv], "y": y[inv], "x": x[inv], "state": state[inv]}, int(steps.value), el


_PAD_N_FACES = {"low": 0, "high": 0, "none": -1, "both": +1}  # _sgrid/core.py:41-49 (n_faces - n_nodes)


def grid_meta(case: dict) -> dict:
    """Axes, ravel dims and C-grid offsets exactly as XGrid derives them (xgrid.py:21-24,137-139,208-231,
    _xinterpolators.py:99-109) from the case's dimension names."""
    lon = np.asarray(case["lon"])
    depth = case.get("depth")
    nodes = {"XG": lon.shape[-1], "YG": (np.asarray(case["lat"]).shape[0])}
    if depth is not None:
        nodes["depth"] = len(depth)
    sizes = dict(nodes)
    for name, dims in case.get("field_dims", {}).items():
        shp = np.asarray(case["fields"][name]).shape
        for d, s in zip(dims, shp):
            if d in ("XC", "YC", "ZC"):
                sizes[d] = s
    pads = {"X": case.get("x_pad", "low"), "Y": case.get("y_pad", "low"), "Z": case.get("z_pad", "both")}
    names = {"X": ("XC", "XG"), "Y": ("YC", "YG"), "Z": ("ZC", "depth")}
    out = {}
    for ax in "XYZ":
        face, node = names[ax]
        present = (face in sizes or node in sizes) and not (ax == "Z" and depth is None)
        out["has_" + ax.lower()] = int(present)
        if not present:
            out[ax.lower() + "dim"] = 0
        elif face in sizes:
            out[ax.lower() + "dim"] = sizes[face] - 1
        else:
            out[ax.lower() + "dim"] = sizes[node] + _PAD_N_FACES[pads[ax]] - 1
        out["off_" + ax.lower()] = 1 if pads[ax] == "low" else 0
    if depth is None:
        out["off_z"] = 0
    return out


def _native_widths(arr):
    a = np.asarray(arr)
    if a.ndim != 1 or a.shape[0] < 2:
        return None
    return np.ascontiguousarray((a[1:] - a[:-1]).astype(np.float64))


class MarshalledCase:
    """Keeps every NumPy buffer alive while the C structs point into them."""

    def __init__(self, case: dict, hash_table: dict | None = None):
        self.case = case
        self.keep = []
        mesh = case["mesh"]
        spherical = mesh == "spherical"
        deg2m = EARTH_RADIUS * np.pi / 180.0 if spherical else 1.0
        lon = np.asarray(case["lon"])
        lat = np.asarray(case["lat"])
        depth = case.get("depth")
        meta = grid_meta(case)
        g = PoGrid()
        g.kind = 1 if lon.ndim == 2 else 0
        g.spherical = int(spherical)
        g.has_x, g.has_y, g.has_z = meta["has_x"], meta["has_y"], meta["has_z"]
        g.nx = lon.shape[-1]
        g.ny = lat.shape[0]
        g.nz = len(depth) if depth is not None else 0
        g.xdim, g.ydim, g.zdim = meta["xdim"], meta["ydim"], meta["zdim"]
        g.off_x, g.off_y, g.off_z = meta["off_x"], meta["off_y"], meta["off_z"]
        g.lon_f32 = int(lon.dtype == np.float32)
        g.lat_f32 = int(lat.dtype == np.float32)
        g.depth_f32 = int(depth is not None and np.asarray(depth).dtype == np.float32)
        g.deg2m = deg2m
        lon64 = np.ascontiguousarray(lon, dtype=np.float64)
        lat64 = np.ascontiguousarray(lat, dtype=np.float64)
        dep64 = np.ascontiguousarray(depth, dtype=np.float64) if depth is not None else None
        dlon, dlat, ddep = _native_widths(lon), _native_widths(lat), (_native_widths(depth) if depth is not None else None)
        self.keep += [lon64, lat64, dep64, dlon, dlat, ddep]
        g.lon, g.lat, g.depth = _ptr(lon64), _ptr(lat64), _ptr(dep64)
        g.dlon, g.dlat, g.ddepth = _ptr(dlon), _ptr(dlat), _ptr(ddep)
        if g.kind == 1:
            if hash_table is None:
                hash_table = case.get("hash_table")
            if hash_table is None:
                raise ValueError("curvilinear case needs a spatial-hash table")
            keys = np.ascontiguousarray(hash_table["keys"], dtype=np.uint32)
            starts = np.ascontiguousarray(hash_table["starts"], dtype=np.int64)
            counts = np.ascontiguousarray(hash_table["counts"], dtype=np.int64)
            faces = np.ascontiguousarray(hash_table["faces"], dtype=np.uint32)
            self.keep += [keys, starts, counts, faces]
            g.h_keys, g.h_starts, g.h_counts, g.h_faces = _ptr(keys), _ptr(starts), _ptr(counts), _ptr(faces)
            g.h_nkeys = len(keys)
            g.h_bitwidth = int(hash_table["bitwidth"])
            for i, v in enumerate(np.asarray(hash_table["bbox"], dtype=np.float64)):
                g.h_bbox[i] = float(v)
        grids = [g]

        constants = case.get("constants") or {}
        if constants:
            cg = PoGrid()  # the 1x1 constant-field grid (model.py:292-317): axes X,Y only, ravel dims 0
            cg.kind = 0
            cg.spherical = int(case.get("const_mesh", "flat") == "spherical")
            cg.has_x, cg.has_y, cg.has_z = 1, 1, 0
            cg.nx = cg.ny = 1
            cg.nz = 0
            cg.deg2m = EARTH_RADIUS * np.pi / 180.0 if cg.spherical else 1.0
            z1 = np.zeros(1)
            self.keep.append(z1)
            cg.lon = cg.lat = _ptr(z1)
            grids.append(cg)
        self.grids = (PoGrid * len(grids))(*grids)

        time_s = case.get("time_s")
        has_ti = time_s is not None and len(time_s) > 1
        time64 = np.ascontiguousarray(time_s, dtype=np.float64) if has_ti else None
        self.keep.append(time64)
        self.field_index = {}
        flds = []
        for name, arr in case["fields"].items():
            a = np.asarray(arr)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float64)
            a = np.ascontiguousarray(a)
            self.keep.append(a)
            dims = case["field_dims"][name]
            f = PoField()
            f.grid = 0
            f.dtype = int(a.dtype == np.float64)
            f.nt, f.nz, f.ny, f.nx = a.shape
            f.has_t = int(dims[0] == "time")
            f.has_z = int(dims[1] in ("depth", "ZC"))
            f.has_y = int(dims[2] in ("YG", "YC"))
            f.has_x = int(dims[3] in ("XG", "XC"))
            f.has_time_interval = int(has_ti and dims[0] == "time")
            f.is_const = SCALAR_INTERP[(case.get("scalar_interp") or {}).get(name, "XLinear")]
            f.data = _ptr(a)
            f.time = _ptr(time64)
            self.field_index[name] = len(flds)
            flds.append(f)
        for name, val in constants.items():
            a = np.full((1, 1, 1, 1), val, dtype=np.float64)
            self.keep.append(a)
            f = PoField()
            f.grid = 1
            f.dtype = 1
            f.nt = f.nz = f.ny = f.nx = 1
            f.has_t = f.has_z = 0
            f.has_y = f.has_x = 1
            f.has_time_interval = 0
            f.is_const = 1
            f.data = _ptr(a)
            self.field_index[name] = len(flds)
            flds.append(f)
        self.fields = (PoField * len(flds))(*flds)
        self.ngrids = len(grids)

    def params(self, *, kernels, endtime, dt0, context=None, seed=0, have_guess0=0) -> PoParams:
        case = self.case
        context = dict(context or {})
        p = PoParams()
        p.nk = len(kernels)
        samples = case.get("sample_into") or {}
        self.sample_vars = []
        for i, k in enumerate(kernels):
            p.sample_field[i] = p.sample_var[i] = -1
            if k in samples:  # the user kernel `particles.<var> = fieldset.<F>[particles]`
                fname, vname, _ = samples[k]
                cols = []
                for vn in sample_names(vname):
                    if vn is None:
                        cols.append(0xFF)
                        continue
                    if vn not in self.sample_vars:
                        self.sample_vars.append(vn)
                    cols.append(self.sample_vars.index(vn))
                p.kernels[i] = KERNEL_IDS["SampleField"]
                p.sample_field[i] = {"UV": -2, "UVW": -3}[fname] if fname in ("UV", "UVW") else self.field_index[fname]
                p.sample_var[i] = sum(c << (8 * j) for j, c in enumerate(cols)) if fname in ("UV", "UVW") else cols[0]
            else:
                p.kernels[i] = KERNEL_IDS[k]
        p.cgrid = {"free": 2, "partial": 3}.get(case.get("slip"), int(bool(case.get("cgrid"))))
        p.rk45_mode = int("RK45_tol" in context)
        p.have_guess0 = int(have_guess0)
        p.next_dt_f32 = int(np.dtype(case.get("next_dt_dtype", "float64")) == np.float32)
        fi = self.field_index
        p.fU, p.fV, p.fW = fi.get("U", -1), fi.get("V", -1), fi.get("W", -1)
        p.fKhz, p.fKhm = fi.get("Kh_zonal", -1), fi.get("Kh_meridional", -1)
        p.endtime = float(endtime)
        p.dt0 = float(dt0)
        p.rk45_tol = float(context.get("RK45_tol", 0.0))
        p.rk45_min_dt = float(context.get("RK45_min_dt", 0.0))
        p.rk45_max_dt = float(context.get("RK45_max_dt", 0.0))
        p.dres = float(context.get("dres", 0.0))
        p.seed = int(seed)
        return p


def rk45_context_defaults(case: dict) -> dict:
    """What Kernel.check_fieldsets_in_kernels does to fieldset.context (kernel.py:134-159)."""
    ctx = dict(case.get("context") or {})
    if "AdvectionRK45" in case["kernels"]:
        ctx.setdefault("RK45_tol", 10)
        if case["mesh"] == "spherical":
            ctx["RK45_tol"] = ctx["RK45_tol"] / (EARTH_RADIUS * np.pi / 180.0)
        ctx.setdefault("RK45_min_dt", 1)
        ctx.setdefault("RK45_max_dt", 60 * 60 * 24)
    return ctx


def initial_particles(case: dict, ngrids: int) -> dict:
    """SoA dict as ParticleSet.__init__ builds it (particleset.py:59-137, particle.py:182-222)."""
    sdt = np.dtype(case.get("spatial_dtype", "float64"))
    x = np.atleast_1d(np.asarray(case["x"])).astype(sdt)
    n = x.shape[0]
    y = np.atleast_1d(np.asarray(case["y"])).astype(sdt)
    if case.get("z") is None:
        depth = case.get("depth")
        zval = 0.0
        if depth is not None:  # particleset.py:82-93: depth level with the smallest |depth|
            d = np.asarray(depth, dtype=float)
            zval = d[np.argmin(np.abs(d))]
        z = np.full(n, zval).astype(sdt)
    else:
        z = np.atleast_1d(np.asarray(case["z"])).astype(sdt)
        if z.shape[0] == 1 and n > 1:
            z = np.repeat(z, n)
    t0 = case.get("t0")
    t = np.zeros(n) if t0 is None else np.broadcast_to(np.asarray(t0, dtype=np.float64), (n,)).copy()
    d = {
        "t": t.astype(np.float64),
        "z": z,
        "y": y,
        "x": x,
        "dz": np.zeros(n, sdt),
        "dy": np.zeros(n, sdt),
        "dx": np.zeros(n, sdt),
        "particle_id": np.arange(n, dtype=np.int64),
        "dt": np.ones(n, np.float64),
        "state": np.full(n, 10, np.int32),
        "ei": np.zeros((n, ngrids), np.int32),
    }
    if "AdvectionRK45" in case["kernels"]:
        d["next_dt"] = np.full(n, float(case.get("next_dt0", case["dt"])), np.dtype(case.get("next_dt_dtype", "float64")))
    for fname, vname, vdt in (case.get("sample_into") or {}).values():
        for vn in sample_names(vname):
            if vn is not None:
                d[vn] = np.zeros(n, np.dtype(vdt))
    return d


def sample_names(vname):
    """Variable names of a `sample_into` entry: one name (scalar field) or a list with None for discarded vector components."""
    return list(vname) if isinstance(vname, (list, tuple)) else [vname]


def execute(mc: MarshalledCase, data: dict, *, kernels, endtime, dt0, context=None, seed=0, have_guess0=0, nthreads=1, batch_stop=True,
            _max_iters=0, call_wide_time_error=True, _twe_key=()):
    """One Kernel.execute(pset, endtime, dt) call on the SoA dict ``data`` (updated in place).

    Deleted particles are compacted afterwards like ``Kernel.remove_deleted`` (kernel.py:98-106).  ``batch_stop``: the reference
    checks the error codes after every iteration of its batch loop (kernel.py:236-245), so when some particle errs in its k-th
    iteration, every particle has made at most k iterations when the exception is raised: the per-particle C loop reports that k
    and the call is run again from the same inputs with ``max_iters = k`` (False: every other particle runs on to ``endtime``).
    ``call_wide_time_error`` (the reference's behaviour and the default; False = every particle on its own): a sample outside a field's time
    interval fails the whole call in the reference -- every particle evaluated in that iteration takes code 70 and the value 0 at that
    sample (index_search.py:85-86, field.py:31-44); a first pass finds the first such sample, the call then runs with it.
    """
    if call_wide_time_error:
        saved = {k: np.array(v, copy=True) for k, v in data.items()}
        keys = []
        while len(keys) < PO_MAX_TWE:  # every pass applies the events found so far and reports the next sample at which somebody leaves a time interval
            st = execute(mc, data, kernels=kernels, endtime=endtime, dt0=dt0, context=context, seed=seed, have_guess0=have_guess0,
                         nthreads=nthreads, batch_stop=batch_stop, call_wide_time_error=False, _twe_key=tuple(keys))
            if not st["first_time_error_key"]:
                return st
            keys.append(st["first_time_error_key"])
            for k in list(data):
                data[k] = np.array(saved[k], copy=True)
        raise RuntimeError(f"more than {PO_MAX_TWE} call-wide time errors in one Kernel.execute")
    if batch_stop:
        saved = {k: np.array(v, copy=True) for k, v in data.items()}
        st = execute(mc, data, kernels=kernels, endtime=endtime, dt0=dt0, context=context, seed=seed, have_guess0=have_guess0,
                     nthreads=nthreads, batch_stop=False, call_wide_time_error=False, _twe_key=_twe_key)
        if not st["first_error_iter"]:
            return st
        for k in list(data):
            data[k] = saved[k]
        return execute(mc, data, kernels=kernels, endtime=endtime, dt0=dt0, context=context, seed=seed, have_guess0=have_guess0,
                       nthreads=nthreads, batch_stop=False, call_wide_time_error=False, _max_iters=st["first_error_iter"], _twe_key=_twe_key)
    n = data["x"].shape[0]
    sdt = data["x"].dtype
    w = {k: np.ascontiguousarray(data[k], dtype=np.float64) for k in ("t", "z", "y", "x", "dz", "dy", "dx", "dt")}
    nd = np.ascontiguousarray(data["next_dt"], dtype=np.float64) if "next_dt" in data else None
    state = np.ascontiguousarray(data["state"], dtype=np.int32)
    ei = np.ascontiguousarray(data["ei"], dtype=np.int32)
    pid = np.ascontiguousarray(data["particle_id"], dtype=np.int64)
    P = PoParticles()
    P.n = n
    P.ngrids = ei.shape[1]
    P.spatial_f32 = int(sdt == np.float32)
    P.t, P.z, P.y, P.x = _ptr(w["t"]), _ptr(w["z"]), _ptr(w["y"]), _ptr(w["x"])
    P.dz, P.dy, P.dx, P.dt = _ptr(w["dz"]), _ptr(w["dy"]), _ptr(w["dx"]), _ptr(w["dt"])
    P.next_dt = _ptr(nd)
    P.state, P.ei, P.particle_id = _ptr(state), _ptr(ei), _ptr(pid)
    prm = mc.params(kernels=kernels, endtime=endtime, dt0=dt0, context=context, seed=seed, have_guess0=have_guess0)
    prm.max_iters = int(_max_iters)
    for k, key in enumerate(tuple(_twe_key)[:PO_MAX_TWE]):
        prm.twe_key[k] = int(key)
    xw = {}
    for k, vname in enumerate(getattr(mc, "sample_vars", [])):
        xw[vname] = np.ascontiguousarray(data[vname], dtype=np.float64)
        P.extra[k] = xw[vname].ctypes.data
        P.extra_f32[k] = int(data[vname].dtype == np.float32)
    st = PoStats()
    rc = lib().po_execute(mc.grids, C.c_int32(mc.ngrids), mc.fields, C.c_int32(len(mc.fields)), C.byref(prm), C.byref(P),
                          C.byref(st), C.c_int32(nthreads))
    assert rc == 0
    for k in ("z", "y", "x", "dz", "dy", "dx"):
        data[k] = w[k].astype(sdt)
    data["t"], data["dt"] = w["t"], w["dt"]
    if nd is not None:
        data["next_dt"] = nd.astype(data["next_dt"].dtype)  # exact: the C code already rounded f32 columns
    data["state"], data["ei"], data["particle_id"] = state, ei, pid
    for vname, w_ in xw.items():
        data[vname] = w_.astype(data[vname].dtype)  # exact: the C code already rounded float32 Variables
    keep = data["state"] != 30
    if not keep.all():
        for k in list(data):
            data[k] = data[k][keep]
    return {"steps": st.steps, "attempts": st.attempts, "first_error_iter": int(st.first_error_iter),
            "first_time_error_key": int(st.first_time_error_key)}


def populate_indices(mc: MarshalledCase, data: dict):
    """ei[:, 0] = ravel(grid.search(z, y, x)) with no guess (particleset.py:252-262)."""
    g = mc.grids[0]
    n = data["x"].shape[0]
    x = np.ascontiguousarray(data["x"], dtype=np.float64)
    y = np.ascontiguousarray(data["y"], dtype=np.float64)
    z = np.ascontiguousarray(data["z"], dtype=np.float64)
    L = lib()
    zi = np.zeros(n, np.int32)
    bc = np.zeros(n)
    if g.has_z:
        dep = np.ascontiguousarray(mc.case["depth"], dtype=np.float64)
        L.po_search_1d(_ptr(dep), C.c_int32(len(dep)), _ptr(z), C.c_int64(n), _ptr(zi), _ptr(bc))
    yi = np.zeros(n, np.int32)
    xi = np.zeros(n, np.int32)
    if g.kind == 1:
        xs = np.zeros(n)
        et = np.zeros(n)
        L.po_hash_query(C.byref(g), C.c_int64(n), _ptr(y), _ptr(x), _ptr(yi), _ptr(xi), _ptr(xs), _ptr(et))
    else:
        lon = np.ascontiguousarray(mc.case["lon"], dtype=np.float64)
        lat = np.ascontiguousarray(mc.case["lat"], dtype=np.float64)
        L.po_search_1d(_ptr(lat), C.c_int32(len(lat)), _ptr(y), C.c_int64(n), _ptr(yi), _ptr(bc))
        L.po_search_1d(_ptr(lon), C.c_int32(len(lon)), _ptr(x), C.c_int64(n), _ptr(xi), _ptr(bc))
    ei = np.zeros(n, np.int64)
    stride = 1
    for has, dim, idx in ((g.has_x, g.xdim, xi), (g.has_y, g.ydim, yi), (g.has_z, g.zdim, zi)):
        if has:
            ei += idx.astype(np.int64) * stride
            stride *= dim
    data["ei"][:, 0] = ei.astype(np.int32)


def sample_case(case: dict):
    """Field.eval at explicit points through po_eval (scalar field ``case['sample_field']``)."""
    mc = MarshalledCase(case)
    fidx = mc.field_index[case["sample_field"]]
    t, z, y, x = (np.ascontiguousarray(case[k], dtype=np.float64) for k in ("t0", "z", "y", "x"))
    m = x.shape[0]
    out = np.zeros(m)
    st = np.zeros(m, np.int32)
    prm = mc.params(kernels=[], endtime=0.0, dt0=1.0)
    rc = lib().po_eval(mc.grids, mc.fields, C.byref(prm), C.c_int32(fidx), C.c_int64(m), _ptr(t), _ptr(z), _ptr(y), _ptr(x), _ptr(out),
                       None, None, _ptr(st))
    assert rc == 0
    return {"value": out, "state": st}


ERRORS_TO_THROW = [  # kernel.py:31-38 (order matters)
    (70, "OutsideTimeInterval"),
    (60, "FieldOutOfBoundError"),
    (61, "FieldOutOfBoundSurfaceError"),
    (51, "FieldInterpolationError"),
    (52, "GridSearchingError"),
    (50, "GeneralError"),
]


def run_case(case: dict, nthreads: int = 1, call_wide_time_error: bool = True):
    """ParticleSet.execute (particleset.py:355-470): one Kernel.execute per output interval to the end time -- and, with case["more_calls"]
    ([{"dt": .., "runtime": ..}, ...]), further execute() calls on the same set, as a script loop makes them: every call sets `dt` anew
    (:381), takes its start time from the particles (:523-585) and constructs a Kernel -- which on a spherical mesh divides RK45_tol by deg2m
    AGAIN (kernel.py:144-145 writes the converted value back into the context)."""
    mc = MarshalledCase(case)
    data = initial_particles(case, mc.ngrids)
    ctx = dict(case.get("context") or {})
    have_guess0 = 0
    if case.get("populate"):  # ParticleSet.populate_indices (particleset.py:252-262)
        populate_indices(mc, data)
        have_guess0 = 1
    stats = {}
    obs = []
    calls = [dict(dt=float(case["dt"]), runtime=case.get("runtime"), endtime=case.get("endtime"), outputdt=case.get("outputdt"))]
    calls += [dict(dt=float(c["dt"]), runtime=float(c["runtime"]), endtime=None, outputdt=None) for c in (case.get("more_calls") or ())]
    for call in calls:
        if len(data["state"]) == 0 and not call["outputdt"]:
            break  # particleset.py:366: `if len(self) == 0: return`
        ctx = rk45_context_defaults(dict(case, context=ctx))  # (one Kernel construction per execute)
        if call is not calls[0]:  # np.any(xi) over the guesses the previous call left (index_search.py:269), like parcels_amd.Kernel._have_guess0
            g0 = mc.grids[0]
            have_guess0 = int(bool(g0.kind == 1 and g0.has_x and np.any(np.mod(data["ei"][:, 0].astype(np.int64), max(int(g0.xdim), 1)) != 0)))
        dt = call["dt"]
        sign = 1 if dt > 0 else -1
        data["dt"][:] = dt
        first = data["t"].min() if sign == 1 else data["t"].max()
        start = first
        if call["endtime"] is not None:
            end = float(call["endtime"])
        else:
            end = start + sign * float(call["runtime"])
        # output intervals (particleset.py:440-462): one Kernel.execute per interval, dt is NOT reset in between
        stops = [end]
        if call["outputdt"]:  # next_output accumulates (particleset.py:441,455): k * outputdt would round differently
            stops = []
            next_output = start + float(call["outputdt"]) * sign
            time = start
            while sign * (time - end) < 0:
                time = (min if sign > 0 else max)(next_output, end)
                stops.append(time)
                if abs(time - next_output) < 0.001:
                    next_output += float(call["outputdt"]) * sign
        # what the loop hands a ParticleFile (particleset.py:401-403, 436-457): the whole set at the start and at every output time
        snap = lambda tm: obs.append((float(tm), {k: np.array(data[k], copy=True) for k in ("particle_id", "t", "z", "y", "x")}))  # noqa: E731
        next_output = None
        if call["outputdt"]:
            snap(start)
            next_output = start + float(call["outputdt"]) * sign
        stop_all = False
        for stop in stops:
            if len(data["state"]) > 0:  # (an emptied set: the reference's loop goes on to the end time, writing empty tables, particleset.py:444-462)
                stats = execute(mc, data, kernels=case["kernels"], endtime=stop, dt0=dt, context=ctx, seed=case.get("seed", 0),
                                have_guess0=have_guess0, nthreads=nthreads, call_wide_time_error=call_wide_time_error)
                have_guess0 = 1
            if np.any(data["state"] >= 50):
                stop_all = True
                break
            if len(data["state"]) == 0 and next_output is None:
                break
            if next_output is not None and abs(stop - next_output) < 0.001:
                snap(next_output)
                next_output += float(call["outputdt"]) * sign
        if stop_all:
            break
    if obs:
        stats = dict(stats)
        stats["observations"] = obs
    err = None
    for code, name in ERRORS_TO_THROW:
        if np.any(data["state"] == code):
            err = name
            break
    return data, err, stats
