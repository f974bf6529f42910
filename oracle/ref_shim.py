"""TEST INFRASTRUCTURE ONLY -- loader that runs the *unmodified* reference hot path in this container.

The reference (Parcels v4-alpha, /root/reference, read-only) cannot be imported as a package here
(needs Python >= 3.11 and xarray/dask/zarr/... which are absent).  Its hot-path
modules however only need a pointwise ``DataArray.isel`` gather (``_xinterpolators.py:73-75``), so this
file registers ~60 lines of stub modules and then imports the reference's own

    _core/{statuscodes, mesh, index_search, spatialhash, basegrid, particle, particlesetview, kernel,
           field, particleset, xgrid}, interpolators/_xinterpolators, kernels/*, _sgrid

straight from ``/root/reference/src/parcels`` without touching them.  Nothing here is product code and
nothing is copied from the reference: the functions that run are the reference's own.

Used by ``oracle/make_golden.py`` to generate ``tests/golden/*.npz`` (the reference cannot travel to the
GPU box, the fixtures can) and by ``tests/test_oracle_vs_reference.py`` (skipped when /root/reference is
absent) to pin ``oracle/parcels_oracle.c`` against the real reference.
"""

from __future__ import annotations

import os
import sys
import types
import typing

import numpy as np

REFERENCE_SRC = os.environ.get("PARCELS_REFERENCE_SRC", "/root/reference/src")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "parcels", "_core"))


class _Any:
    """Permissive placeholder for attributes of stubbed third-party modules that are never exercised."""

    def __getattr__(self, name):
        return _Any()

    def __call__(self, *a, **k):
        # decorator-safe: @stub.decorator on a function returns the function
        return a[0] if (len(a) == 1 and callable(a[0]) and not k) else _Any()

    def __or__(self, other):
        return self

    def __ror__(self, other):
        return self

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


def _stub(name: str, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    mod.__getattr__ = lambda n: _Any()  # type: ignore[assignment]
    sys.modules[name] = mod
    return mod


class DA:
    """Minimal xarray.DataArray stand-in: ``.data/.values/.dims/.shape/.ndim`` and pointwise ``isel``."""

    def __init__(self, data, dims=None, coords=None, **k):
        self.data = np.asarray(data)
        self.dims = (dims,) if isinstance(dims, str) else tuple(dims or ())
        self.coords = dict(coords or {})

    @property
    def values(self):
        return self.data

    @property
    def shape(self):
        return self.data.shape

    @property
    def ndim(self):
        return self.data.ndim

    def __len__(self):
        return len(self.data)

    def __getitem__(self, key):
        return DA(self.data[key])

    def isel(self, sel):
        # vectorised pointwise gather over a shared "points" dimension
        # (what xarray does for DataArray indexers that share a dim; pinned by
        #  the reference's tests/test_interpolation.py:208-277)
        idx = tuple(sel[d].data if d in sel else slice(None) for d in self.dims)
        # dims that are not selected are the size-1 "mock" axes (xgrid.py:71-105); the caller reshapes
        # the flat result (``_xinterpolators.py:75``), so their position in the output is irrelevant
        return DA(self.data[idx], dims=("points",))

    def __getattr__(self, name):
        coords = self.__dict__.get("coords", {})
        if name in coords:
            return coords[name]
        raise AttributeError(name)


class _DS:
    pass


_LOADED: dict | None = None


def load_reference() -> dict:
    """Install the stubs (once) and import the reference's hot-path modules. Returns a name->module dict."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not reference_available():
        raise RuntimeError(f"reference sources not found under {REFERENCE_SRC}")

    if not hasattr(typing, "Self"):
        typing.Self = typing.Any  # type: ignore[attr-defined]  # py3.10

    if "parcels" in sys.modules and not getattr(sys.modules["parcels"], "_is_ref_shim", False):
        raise RuntimeError("a real `parcels` package is already imported; cannot install the reference shim")
    pkg = types.ModuleType("parcels")
    pkg.__path__ = [os.path.join(REFERENCE_SRC, "parcels")]  # parcels/__init__.py is never executed
    pkg._is_ref_shim = True
    sys.modules["parcels"] = pkg

    if "xarray" not in sys.modules:
        _stub("xarray", DataArray=DA, Dataset=_DS)
    if "cftime" not in sys.modules:
        _stub("cftime", datetime=type("datetime", (), {}))
    if "dask" not in sys.modules:
        _stub("dask", is_dask_collection=lambda x: False)
        _stub("dask.base", is_dask_collection=lambda x: False)
        _stub("dask.array")
    for name in [
        "uxarray",
        "zarr",
        "zarr.storage",
        "zarr.abc",
        "zarr.abc.store",
        "cf_xarray",
        "polars",
        "pooch",
        "netCDF4",
        "xgcm",
    ]:
        if name not in sys.modules:
            _stub(name)
    _stub("parcels._repr_utils")

    import importlib

    sys.dont_write_bytecode = True  # never write .pyc files into the read-only reference tree

    names = {
        "statuscodes": "parcels._core.statuscodes",
        "mesh": "parcels._core.mesh",
        "time": "parcels._core.utils.time",
        "interp_utils": "parcels._core.utils.interpolation",
        "index_search": "parcels._core.index_search",
        "spatialhash": "parcels._core.spatialhash",
        "basegrid": "parcels._core.basegrid",
        "particle": "parcels._core.particle",
        "particlesetview": "parcels._core.particlesetview",
        "kernel": "parcels._core.kernel",
        "field": "parcels._core.field",
        "particleset": "parcels._core.particleset",
        "xgrid": "parcels._core.xgrid",
        "xinterp": "parcels.interpolators._xinterpolators",
        "kernels": "parcels.kernels",
        "sgrid": "parcels._sgrid",
    }
    mods = {k: importlib.import_module(v) for k, v in names.items()}
    _LOADED = mods
    return mods


# ----------------------------------------------------------------------------------------------------
# Wiring a reference FieldSet without model.py / fieldset.py (those need real xarray; neither is on the
# arithmetic path).  Everything below only *assembles* reference objects.
# ----------------------------------------------------------------------------------------------------


class _FakeDs:
    """dict-like dataset exposing what XGrid touches: .dims, .sizes, lon/lat/depth as attribute and item."""

    def __init__(self, sizes: dict, coords: dict):
        self.sizes = dict(sizes)
        self.dims = set(sizes)
        self._coords = coords

    def __getitem__(self, k):
        return self._coords[k]

    def __contains__(self, k):
        return k in self._coords

    def __getattr__(self, k):
        c = self.__dict__.get("_coords", {})
        if k in c:
            return c[k]
        raise AttributeError(k)


class _FakeModel:
    def __init__(self, grid, data, time_interval):
        self.grid = grid
        self.data = data
        self._time_interval = time_interval
        self.field_to_interpolator = {}

    def field_data(self, name):
        return self.data[name]

    @property
    def time_interval(self):
        return self._time_interval


class RefFieldSet:
    """Duck-typed stand-in for the reference FieldSet (fieldset.py:37-222): fields/gridset/context/time_interval."""

    def __init__(self, fields: dict, gridset: list, time_interval):
        object.__setattr__(self, "fields", fields)
        object.__setattr__(self, "gridset", gridset)
        object.__setattr__(self, "time_interval", time_interval)
        object.__setattr__(self, "context", {})

    def add_context(self, name, value):
        self.context[name] = value

    def __getattr__(self, name):
        d = self.__dict__
        if name in d.get("fields", {}):
            return d["fields"][name]
        if name in d.get("context", {}):
            return d["context"][name]
        raise AttributeError(name)  # kernel.py:118 relies on hasattr() being False for unknown names


def make_ref_grid(*, lon, lat, depth, mesh, x_pad="low", y_pad="low", z_pad="both", sizes_extra=None):
    """Instantiate the reference XGrid (object.__new__, no xarray) for 1-D or 2-D lon/lat.

    Dimension naming convention used by all our reference wiring:
      nodes  : XG, YG, depth      faces : XC, YC, ZC
    ``sizes_extra`` may add face dims (e.g. {"XC": nx}) when fields live on them.
    """
    m = load_reference()
    sgrid, XGrid = m["sgrid"], m["xgrid"].XGrid
    pad = {"low": sgrid.Padding.LOW, "high": sgrid.Padding.HIGH, "both": sgrid.Padding.BOTH, "none": sgrid.Padding.NONE}
    lon = np.asarray(lon)
    lat = np.asarray(lat)
    has_z = depth is not None
    meta = sgrid.SGrid2DMetadata(
        cf_role="grid_topology",
        topology_dimension=2,
        node_dimensions=("XG", "YG"),
        node_coordinates=("lon", "lat"),
        face_dimensions=(
            sgrid.FaceNodePadding("XC", "XG", pad[x_pad]),
            sgrid.FaceNodePadding("YC", "YG", pad[y_pad]),
        ),
        vertical_dimensions=(sgrid.FaceNodePadding("ZC", "depth", pad[z_pad]),) if has_z else None,
    )
    if lon.ndim == 1:
        sizes = {"XG": lon.shape[0], "YG": lat.shape[0]}
        coords = {"lon": DA(lon, dims=("XG",)), "lat": DA(lat, dims=("YG",))}
    else:
        sizes = {"XG": lon.shape[1], "YG": lon.shape[0]}
        coords = {"lon": DA(lon, dims=("YG", "XG")), "lat": DA(lat, dims=("YG", "XG"))}
    if has_z:
        depth = np.asarray(depth)
        if not np.issubdtype(depth.dtype, np.floating):
            depth = depth.astype(float)
        sizes["depth"] = depth.shape[0]
        coords["depth"] = DA(depth, dims=("depth",))
    sizes.update(sizes_extra or {})
    g = object.__new__(XGrid)
    g.sgrid_metadata = meta
    g._mesh = m["mesh"].get_mesh(mesh)
    g._spatialhash = None
    g._ds = _FakeDs(sizes, coords)
    return g


def make_ref_fieldset(*, grid, fields: dict, time_s=None, cgrid=False, constants=None, const_mesh="flat", slip=None):
    """Assemble reference Field/VectorField objects.

    fields : name -> (ndarray TZYX, dims tuple of 4 names).  Use "mockT"/"mockZ" style names for
             size-1 axes the field does not have (mirrors xgrid.py:71-105).
    time_s : 1-D float seconds of the time levels (None => no time_interval, time-invariant fields)
    """
    m = load_reference()
    Field, VectorField = m["field"].Field, m["field"].VectorField
    xi = m["xinterp"]
    TimeInterval = m["time"].TimeInterval

    tint = None
    tcoord = None
    if time_s is not None and len(time_s) > 1:
        ts = np.asarray(time_s, dtype=float)
        t64 = (ts * 1e9).round().astype("int64").astype("timedelta64[ns]")
        tint = TimeInterval(t64[0], t64[-1])
        tcoord = DA(t64, dims=("time",))
    data = {}
    for name, (arr, dims) in fields.items():
        da = DA(np.asarray(arr), dims=tuple(dims))
        if tcoord is not None and "time" in dims:
            da.coords = {"time": tcoord}
        data[name] = da
    model = _FakeModel(grid, data, tint)
    fobjs: dict = {}
    for name in data:
        f = Field(name, model)
        f.igrid = 0
        f.interp_method = xi.XLinear()
        fobjs[name] = f
    vinterp = xi.CGrid_Velocity if cgrid else xi.XLinear_Velocity
    if slip is not None:
        vinterp = {"free": xi.XFreeslip, "partial": xi.XPartialslip}[slip]
    if "U" in fobjs and "V" in fobjs:
        fobjs["UV"] = VectorField("UV", fobjs["U"], fobjs["V"], interp_method=vinterp())
        if "W" in fobjs:
            fobjs["UVW"] = VectorField("UVW", fobjs["U"], fobjs["V"], fobjs["W"], interp_method=vinterp())
    gridset = [grid]
    if constants:
        cgridobj = make_ref_grid(lon=np.zeros(1), lat=np.zeros(1), depth=None, mesh=const_mesh)
        # constant-field grid of the reference (model.py:292-317): 1x1 nodes, axes X,Y only
        cmodel = _FakeModel(cgridobj, {}, None)
        for name, val in constants.items():
            cmodel.data[name] = DA(np.full((1, 1, 1, 1), val), dims=("mockT", "mockZ", "YG", "XG"))
            f = Field(name, cmodel)
            f.igrid = 1
            f.interp_method = xi.XConstantField()
            fobjs[name] = f
        gridset.append(cgridobj)
    return RefFieldSet(fobjs, gridset, tint)


class _OutputStub:
    """What ParticleSet.execute touches of a ParticleFile (particleset.py:401-403,419-462): it splits the run into
    output intervals, i.e. one Kernel.execute per interval."""

    def __init__(self, outputdt_s):
        self.outputdt = outputdt_s
        self.metadata = {}
        self.path = "<memory>"
        self.times = []
        self.obs = []

    def set_metadata(self, mesh):
        pass

    def write(self, pset, time):
        self.times.append(float(time))
        self.obs.append({k: np.array(pset._data[k], copy=True) for k in ("particle_id", "t", "z", "y", "x")})

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def run_reference(fieldset, kernels, *, x, y, z, t=None, dt, runtime=None, endtime_s=None, spatial_dtype=np.float64,
                  extra_vars=None, particle_kwargs=None, populate=False, outputdt=None, more_calls=()):
    """Run the reference's own ParticleSet.execute and return a copy of its SoA dict (+ raised exception name)."""
    m = load_reference()
    P = m["particle"]
    pclass = P.get_default_particle(spatial_dtype)
    if extra_vars:
        pclass = pclass.add_variable([P.Variable(n, dtype=dt_, initial=init) for n, dt_, init in extra_vars])
    n = len(np.atleast_1d(x))
    if t is None:
        tt = np.repeat(np.timedelta64(0, "s"), n)
    else:
        tt = (np.asarray(t, dtype=float) * 1e9).round().astype("int64").astype("timedelta64[ns]")
    pset = m["particleset"].ParticleSet(fieldset, pclass=pclass, x=x, y=y, z=z, t=tt, **(particle_kwargs or {}))
    if populate:
        pset.populate_indices()
    err = None
    kw = {}
    if runtime is not None:
        kw["runtime"] = runtime
    if endtime_s is not None:
        kw["endtime"] = np.timedelta64(int(round(endtime_s * 1e9)), "ns")
    if outputdt is not None:  # particlefile.py needs real pyarrow/xarray: only the loop's use of it is reproduced
        kw["output_file"] = _OutputStub(float(outputdt))
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            pset.execute(kernels, dt=dt, verbose_progress=False, **kw)
            for call in more_calls:  # the usual script loop: further execute() calls on the same ParticleSet ({"dt": .., "runtime": ..})
                pset.execute(kernels, dt=float(call["dt"]), runtime=float(call["runtime"]), verbose_progress=False)
        except Exception as e:  # per-particle error codes surface as exceptions (kernel.py:239-245)
            err = type(e).__name__
    out = {k: np.array(v, copy=True) for k, v in pset._data.items()}
    stub = kw.get("output_file")
    if stub is not None and stub.obs and all(len(o["x"]) == len(stub.obs[0]["x"]) for o in stub.obs):
        # what the ParticleFile would have been handed at every output time (no deletions: rectangular arrays)
        out["obs_time"] = np.array(stub.times)
        for k in ("particle_id", "t", "z", "y", "x"):
            out["obs_" + k] = np.stack([o[k] for o in stub.obs])
    elif stub is not None and stub.obs:
        # deletions between the output times: ragged -- the observations back to back, observation k = rows obs_offsets[k] : obs_offsets[k + 1]
        out["obs_time"] = np.array(stub.times)
        out["obs_offsets"] = np.concatenate([[0], np.cumsum([len(o["x"]) for o in stub.obs])]).astype(np.int64)
        for k in ("particle_id", "t", "z", "y", "x"):
            out["obs_" + k] = np.concatenate([o[k] for o in stub.obs])
    return out, err
