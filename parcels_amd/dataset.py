"""A tiny labelled-array container standing in for the xarray.Dataset the reference ingests.

xarray is not a dependency of the engine: the hot path only needs, per variable, its dimension names and a NumPy
array, plus the SGRID metadata.  ``Dataset.from_xarray`` converts a real SGRID xarray.Dataset when xarray is
installed, so user scripts written against ``FieldSet.from_sgrid_conventions(ds)`` keep working.
"""

from __future__ import annotations

import numpy as np

from .sgrid import FaceNodePadding, Padding, SGrid2DMetadata


class DataArray:
    def __init__(self, dims, data, attrs=None):
        from .sources import is_level_source

        self.dims = (dims,) if isinstance(dims, str) else tuple(dims)
        # a level source (parcels_amd.sources) stays what it is: its levels are read when the device asks for them
        self.data = data if is_level_source(data) else np.asarray(data)
        if len(self.data.shape) != len(self.dims):
            raise ValueError(f"dims {self.dims} do not match array of shape {self.data.shape}")
        self.attrs = dict(attrs or {})

    @property
    def values(self):
        return self.data

    @property
    def shape(self):
        return self.data.shape

    @property
    def ndim(self):
        return len(self.data.shape)


def _as_da(v) -> DataArray:
    if isinstance(v, DataArray):
        return v
    if isinstance(v, (tuple, list)) and len(v) in (2, 3):
        return DataArray(*v)
    raise TypeError("expected (dims, data[, attrs])")


class Dataset:
    """data_vars / coords: name -> (dims, array[, attrs]); sgrid: SGrid2DMetadata."""

    def __init__(self, data_vars=None, coords=None, sgrid: SGrid2DMetadata | None = None, attrs=None):
        self.data_vars = {k: _as_da(v) for k, v in (data_vars or {}).items()}
        self.coords = {k: _as_da(v) for k, v in (coords or {}).items()}
        self.sgrid = sgrid
        self.attrs = dict(attrs or {})

    @property
    def sizes(self) -> dict:
        s: dict = {}
        for da in list(self.data_vars.values()) + list(self.coords.values()):
            for d, n in zip(da.dims, da.shape):
                if d in s and s[d] != n:
                    raise ValueError(f"conflicting sizes for dimension {d!r}: {s[d]} vs {n}")
                s[d] = n
        return s

    @property
    def dims(self):
        return set(self.sizes)

    def __getitem__(self, k):
        if k in self.data_vars:
            return self.data_vars[k]
        return self.coords[k]

    def __setitem__(self, k, v):
        """ds["W"] = (dims, array) or a DataArray: adds / replaces a data variable (a coordinate if the name is one)."""
        (self.coords if k in self.coords else self.data_vars)[k] = _as_da(v)

    def __contains__(self, k):
        return k in self.data_vars or k in self.coords

    def copy(self):
        return Dataset(dict(self.data_vars), dict(self.coords), self.sgrid, dict(self.attrs))

    @classmethod
    def from_xarray(cls, ds):  # pragma: no cover - xarray is optional
        md = ds.sgrid.metadata if hasattr(ds, "sgrid") else None
        if md is None:
            raise ValueError("dataset carries no SGRID metadata")
        pad = {p.value: Padding(p.value) for p in type(md.face_dimensions[0].padding)}
        fd = tuple(FaceNodePadding(f.face, f.node, pad[f.padding.value]) for f in md.face_dimensions)
        vd = None
        if md.vertical_dimensions is not None:
            vd = tuple(FaceNodePadding(f.face, f.node, pad[f.padding.value]) for f in md.vertical_dimensions)
        meta = SGrid2DMetadata(node_dimensions=md.node_dimensions, face_dimensions=fd,
                               node_coordinates=md.node_coordinates, vertical_dimensions=vd)
        skip = {v for v in ds.data_vars if ds[v].attrs.get("cf_role") == "grid_topology"}
        dv = {k: (ds[k].dims, ds[k].values, dict(ds[k].attrs)) for k in ds.data_vars if k not in skip}
        co = {k: (ds[k].dims, ds[k].values, dict(ds[k].attrs)) for k in ds.coords}
        return cls(dv, co, meta, dict(ds.attrs))
