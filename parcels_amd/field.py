"""Field / VectorField (mirrors src/parcels/_core/field.py) and the structured model that owns them
(the part of src/parcels/_core/model.py the hot path needs).

Sampling -- ``fieldset.UV[t, z, y, x]`` / ``field.eval(...)`` -- runs on the GPU through ``pk_eval``
(include/parcels_hip.h); nothing here interpolates on the host.
"""

from __future__ import annotations

import warnings

import numpy as np

from .dataset import DataArray, Dataset
from .interpolators import (
    CGrid_Velocity,
    ScalarInterpolator,
    XFreeslip,
    XPartialslip,
    VectorInterpolator,
    XConstantField,
    XLinear,
    XLinear_Velocity,
)
from .interpolators import CGrid_Tracer, XLinearInvdistLandTracer, XNearest  # noqa: E402
from .statuscodes import StatusCode
from .xgrid import XGrid

_FIELD_DATA_ORDERING = ("T", "Z", "Y", "X")


class FieldEvalWarning(UserWarning):  # _core/warnings.py:31-38
    """Issues during the evaluation of a Field (out-of-bounds indices during interpolation)."""


class TimeInterval:
    """Closed interval [left, right] (utils/time.py:17-91)."""

    def __init__(self, left, right):
        if left >= right:
            raise ValueError(f"Expected left to be strictly less than right, got left={left} and right={right}.")
        self.left = left
        self.right = right

    @property
    def time_length_as_flt(self) -> float:
        return to_seconds(self.right - self.left)

    def __contains__(self, item):
        return self.left <= item <= self.right

    def __eq__(self, other):
        return isinstance(other, TimeInterval) and self.left == other.left and self.right == other.right

    def __repr__(self):
        return f"TimeInterval(left={self.left!r}, right={self.right!r})"

    def intersection(self, other):
        start, end = max(self.left, other.left), min(self.right, other.right)
        return TimeInterval(start, end) if start < end else None

    def get_cf_attrs(self) -> dict:
        """CF attributes of "x seconds from the left edge" (utils/time.py:88-119).  The reference routes calendar dates through
        cftime.datetime(calendar="gregorian"), whose strftime format is "%Y-%m-%d %H:%M:%S" and whose calendar attribute reads
        back as "standard"; cftime-typed (non-standard calendar) intervals are outside the hot path's scope."""
        import datetime as _dt

        left = self.left
        if isinstance(left, np.timedelta64):
            return {"units": "seconds"}
        if isinstance(left, np.datetime64):
            left = left.astype("datetime64[us]").astype(_dt.datetime)
        if isinstance(left, _dt.datetime):
            return {"units": f"seconds since {left.strftime('%Y-%m-%d %H:%M:%S')}", "calendar": "standard"}
        raise NotImplementedError(f"Not implemented for time object {type(left)=!r}")


def to_seconds(dt) -> float | np.ndarray:
    """utils/time.py:197-214 (timedelta_to_float)."""
    import datetime as _dt

    if isinstance(dt, _dt.timedelta):
        return dt.total_seconds()
    if isinstance(dt, np.timedelta64):
        return float(dt / np.timedelta64(1, "s"))
    if hasattr(dt, "dtype") and np.issubdtype(dt.dtype, np.timedelta64):
        return (dt / np.timedelta64(1, "s")).astype(float)
    return float(dt) if np.ndim(dt) == 0 else np.asarray(dt, dtype=float)


def transpose_to_tzyx(da: DataArray, metadata) -> DataArray:
    """xgrid.py:71-105: order TZYX, size-1 "mock" dims for absent axes."""
    d2a = dict(metadata.dim_to_axis())
    d2a["time"] = "T"
    if all(d not in d2a for d in da.dims):
        if da.shape != (1, 1, 1, 1):
            raise ValueError(f"DataArray with dims {da.dims} has no dimension on the grid")
        return DataArray(tuple(f"mock{a}" for a in _FIELD_DATA_ORDERING), da.data, da.attrs)
    unknown = [d for d in da.dims if d not in d2a]
    if unknown:
        raise ValueError(f"DataArray with dims {da.dims} has dimensions {unknown} that are not on the provided grid")
    axes = [d2a[d] for d in da.dims]
    if len(set(axes)) != len(axes):
        raise ValueError(f"two dimensions of {da.dims} lie on the same axis")
    data = da.data
    dims = list(da.dims)
    for ax in _FIELD_DATA_ORDERING:
        if ax not in axes:
            data = data[None]
            dims.insert(0, f"mock{ax}")
            axes.insert(0, ax)
    order = [axes.index(ax) for ax in _FIELD_DATA_ORDERING]
    return DataArray(tuple(dims[i] for i in order), np.transpose(data, order), da.attrs)


class StructuredModelData:
    """Dataset + grid + interpolator registry (model.py:146-250)."""

    def __init__(self, ds: Dataset, mesh, vector_field_components: dict, skip_field_data_validation=False):
        if not isinstance(ds, Dataset):
            raise ValueError(f"Expected `ds` to be a parcels_amd.Dataset. Got {type(ds)}")
        ds = ds.copy()
        md = ds.sgrid
        from .sources import is_level_source

        for name in list(ds.data_vars):
            if is_level_source(ds.data_vars[name].data):  # read level by level later: must already be laid out (time, z, y, x)
                da = ds.data_vars[name]
                if len(da.dims) != 4 or da.dims[0] != "time":
                    raise ValueError(f"level source '{name}' needs dims (time, <z>, <y>, <x>) naming its four axes; got {da.dims}")
                d2a = dict(md.dim_to_axis())
                axes = [d2a.get(d) for d in da.dims[1:]]
                for want, got, d in zip("ZYX", axes, da.dims[1:]):
                    if got not in (want, None) or (got is None and not str(d).startswith("mock")):
                        raise ValueError(f"level source '{name}': dimension '{d}' is not the {want} axis of the grid (use 'mock{want}' for an absent axis)")
                # NaN -> 0 at read time unless validation is skipped: kept HERE (the engine asks for it), the caller's source object is
                # shared by every FieldSet built from it and is left alone
                self.level_fill_nan = getattr(self, "level_fill_nan", {})
                self.level_fill_nan[name] = bool(getattr(da.data, "fill_nan", True)) and not skip_field_data_validation
                continue
            da = transpose_to_tzyx(ds.data_vars[name], md)
            if not skip_field_data_validation and np.issubdtype(da.data.dtype, np.floating):
                if np.isnan(da.data).any():  # model.py:135-143 fillna(0)
                    da = DataArray(da.dims, np.nan_to_num(da.data, nan=0.0), da.attrs)
            ds.data_vars[name] = da
        self.data = ds
        self.grid = XGrid(ds, mesh)
        self.vector_field_components = dict(vector_field_components)
        self.field_to_interpolator: dict = {}
        self._fields = None
        # time axis in float seconds since its first level
        self.time_values = None
        self.time_flt = None
        if "time" in ds.coords and ds.coords["time"].data.size > 1:
            tv = np.asarray(ds.coords["time"].data)
            self.time_values = tv
            if np.issubdtype(tv.dtype, np.datetime64) or np.issubdtype(tv.dtype, np.timedelta64):
                self.time_flt = to_seconds(tv - tv[0])
            else:
                self.time_flt = tv.astype(np.float64) - float(tv[0])
            if not np.all(np.diff(self.time_flt) > 0):
                raise ValueError("time levels must be strictly increasing")

    @property
    def time_interval(self):  # model.py:511-515
        if self.time_values is None:
            return None
        tv = self.time_values
        if np.issubdtype(tv.dtype, np.datetime64) or np.issubdtype(tv.dtype, np.timedelta64):
            return TimeInterval(tv[0], tv[-1])
        return TimeInterval(np.timedelta64(int(round(float(tv[0]) * 1e9)), "ns"), np.timedelta64(int(round(float(tv[-1]) * 1e9)), "ns"))

    def field_data(self, name):
        return self.data.data_vars[name]

    def construct_fields(self):
        single = {name: Field(str(name), self) for name in self.data.data_vars}
        vectors = {}
        for vname, comps in self.vector_field_components.items():
            u, v = self.data.data_vars[comps[0]], self.data.data_vars[comps[1]]
            agrid = set(u.dims) == set(v.dims)  # model.py:505-508
            interp = XLinear_Velocity() if agrid else CGrid_Velocity()
            vectors[vname] = VectorField(vname, *[single[c] for c in comps], interp_method=interp)
        return list({**single, **vectors}.values())


class Field:
    """Scalar field (field.py:47-195)."""

    def __init__(self, name: str, model: StructuredModelData):
        if not isinstance(name, str) or not name.isidentifier():
            raise ValueError(f"Field name has to be a valid Python variable name. Got {name!r}")
        self.name = name
        self.model = model
        self.igrid = -1
        self._fieldset = None

    @property
    def data(self) -> DataArray:
        return self.model.field_data(self.name)

    @property
    def grid(self) -> XGrid:
        return self.model.grid

    @property
    def time_interval(self):
        if "time" not in self.data.dims:
            return None
        return self.model.time_interval

    def __repr__(self):
        return f"Field(name={self.name})"

    @property
    def interp_method(self):
        try:
            return self.model.field_to_interpolator[self.name]
        except KeyError as e:
            raise AttributeError(f"{type(self).__name__} doesn't have an interp_method defined for it.") from e

    @interp_method.setter
    def interp_method(self, value):
        if not isinstance(value, ScalarInterpolator):
            raise ValueError(f"interp_method must be a `ScalarInterpolator` object. Got {type(value)=!r}")
        if not isinstance(value, (XLinear, XConstantField, XNearest, CGrid_Tracer, XLinearInvdistLandTracer)):
            raise NotImplementedError(f"{type(value).__name__} has no HIP implementation")
        self.model.field_to_interpolator[self.name] = value
        if self._fieldset is not None:  # the scalar interpolator code and the C-grid packing are part of the device descriptors
            self._fieldset._engine = None

    def eval(self, t, z, y, x, particles=None):
        """Interpolate in space and time on the GPU (field.py:145-185). Returns the values as float64.  ``particles`` (the view a
        Python kernel received): the particles the sampling fails on get the reference's error codes (field.py:307-378)."""
        if self._fieldset is None:
            raise RuntimeError("Field is not attached to a FieldSet")
        eng = self._fieldset._engine_or_create()
        with _points_dtype(eng, y):
            val = eng.sample(self.name, *_sample_points(t, z, y, x))[0]
        _mark_particles(particles, eng, self, z, y, x)
        return val

    def __getitem__(self, key):
        if self.name in ("U", "V", "W"):  # field.py:134-143
            warnings.warn("Sampling of velocities should normally be done using fieldset.UV or fieldset.UVW object; tread carefully",
                          RuntimeWarning, stacklevel=2)
        return _eval_key(self, key)


def _sample_points(t, z, y, x):
    """Sample coordinates as float64 arrays (the columns of a kernel's `particles` arrive as write-through proxies)."""
    return tuple(np.asarray(v, dtype=np.float64) if not np.isscalar(v) else v for v in (t, z, y, x))


class _points_dtype:
    """Tell the library when the sample points are float32 columns (the default Particle): `np.cos(np.deg2rad(y))` of the velocity
    conversion is then a float32 cosine in the reference (_xinterpolators.py:183-187), as it is inside a fused launch."""

    def __init__(self, eng, y):
        self.eng = eng
        self.f32 = getattr(y, "dtype", None) == np.float32

    def __enter__(self):
        if self.f32:
            self.eng.ctx.check(self.eng.lib.pk_set_option(self.eng.ctx.handle, b"eval_points_f32", 1), "pk_set_option")

    def __exit__(self, *a):
        if self.f32:
            self.eng.ctx.check(self.eng.lib.pk_set_option(self.eng.ctx.handle, b"eval_points_f32", 0), "pk_set_option")


def _unpack_key(key):
    """`field[particles]` / `field[pset]` / `field[t, z, y, x]` / `field[t, z, y, x, particles]` (field.py:187-195, 297-304)."""
    from .hostkernels import HostParticles

    if isinstance(key, HostParticles):
        return key.t, key.z, key.y, key.x, key
    if hasattr(key, "_data"):  # a ParticleSet (or the single-row view pset[i]): every row it holds
        d = key._data
        idx = getattr(key, "_index", slice(None))
        return d["t"][idx], d["z"][idx], d["y"][idx], d["x"][idx], None
    key = tuple(key)
    return key[0], key[1], key[2], key[3], (key[4] if len(key) > 4 else None)


def _eval_key(field, key):
    """field[key] (field.py:187-195, 297-304).  A sample outside the field's time interval is an error of the particles it was taken for; with
    no particles in the key there is nobody to carry it: the reference raises (field.py:31-37), and so does this."""
    from .hostkernels import HostParticles
    from .statuscodes import OutsideTimeInterval

    t, z, y, x, particles = _unpack_key(key)
    try:
        val = field.eval(t, z, y, x, particles)
    except OutsideTimeInterval:
        if not isinstance(particles, HostParticles):
            raise
        # field.py:31-44 (_deal_with_errors): every particle the sample was taken for carries the error, and the sample is 0 -- a Python
        # scalar per component, as there
        particles.state = int(StatusCode.ErrorOutsideTimeInterval)
        ncomp = {"3D": 3, "2D": 2}.get(getattr(field, "vector_type", None), 1)
        return 0 if ncomp == 1 else (0,) * ncomp
    if particles is None and isinstance(key, tuple):
        st = getattr(field._fieldset._engine_or_create(), "last_sample_state", None)
        if st is not None and np.any(np.asarray(st) == int(StatusCode.ErrorOutsideTimeInterval)):
            raise RuntimeError(f"Field {field.name} sampled outside its time interval. Error could not be handled because particles was not part of the Field Sampling.")
    return val


def _mark_particles(particles, eng, field=None, z=None, y=None, x=None):
    """What sampling does to the particles a kernel passed along (field.py:394-405): their `ei` on the field's grid becomes the cell of
    the sample point (_update_particles_ei, :307-317), the points it fails on get the error codes (:327-378)."""
    from .hostkernels import HostParticles, _apply_sample_states
    from .statuscodes import OutsideTimeInterval

    st = getattr(eng, "last_sample_state", None)
    if isinstance(particles, HostParticles) and st is not None and np.any(np.asarray(st) == int(StatusCode.ErrorOutsideTimeInterval)):
        # index_search.py:85-86: ONE point outside the field's time interval fails the whole call, before any `ei` or state of it is
        # written (field.py:394-405); Field.__getitem__ turns that into code 70 for every particle of the view (_eval_key)
        raise OutsideTimeInterval(None, field)
    if field is not None and isinstance(particles, HostParticles) and len(particles._rows) > 0:
        igrid = eng.grids.index(field.grid)
        _, zz, yy, xx = _sample_points(0.0, z, y, x)
        n = len(particles._rows)
        zz, yy, xx = (np.broadcast_to(np.asarray(v, dtype=np.float64), (n,)) for v in (zz, yy, xx))
        particles._data["ei"][particles._rows, igrid] = eng.search(igrid, zz, yy, xx)
    _apply_sample_states(particles, getattr(eng, "last_sample_state", None))


class VectorField:
    """Vector field (field.py:198-304)."""

    def __init__(self, name, U: Field, V: Field, W: Field | None = None, interp_method=None):
        if interp_method is None:
            raise ValueError("interp_method must be provided for VectorField initialization.")
        if not isinstance(interp_method, VectorInterpolator):
            raise ValueError(f"interp_method must be a `VectorInterpolator` object. Got {type(interp_method)=!r}")
        self.name = name
        self.U, self.V, self.W = U, V, W
        self.grid = U.grid
        self.igrid = U.igrid
        tis = [f.time_interval for f in (U, V) + ((W,) if W is not None else ())]
        if any(ti != tis[0] for ti in tis[1:]):
            raise ValueError("Fields must have the same time domain.")
        self.time_interval = U.time_interval
        self.vector_type = "3D" if W is not None else "2D"
        self._interp_method = interp_method
        self._fieldset = None

    @property
    def interp_method(self):
        return self._interp_method

    @interp_method.setter
    def interp_method(self, method):
        if not isinstance(method, VectorInterpolator):
            raise ValueError(f"method must be a `VectorInterpolator` object. Got {type(method)=!r}")
        if not isinstance(method, (XLinear_Velocity, CGrid_Velocity, XFreeslip, XPartialslip)):
            raise NotImplementedError(f"{type(method).__name__} has no HIP implementation yet")
        self._interp_method = method
        if self._fieldset is not None:  # C-grid component packing is decided when the device copy is made
            self._fieldset._engine = None

    def eval(self, t, z, y, x, particles=None):
        if self._fieldset is None:
            raise RuntimeError("VectorField is not attached to a FieldSet")
        eng = self._fieldset._engine_or_create()
        with _points_dtype(eng, y):
            u, v, w = eng.sample(self.name, *_sample_points(t, z, y, x))
        _mark_particles(particles, eng, self.U, z, y, x)
        return (u, v, w) if self.vector_type == "3D" else (u, v)

    def __getitem__(self, key):
        return _eval_key(self, key)
