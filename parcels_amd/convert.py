"""Model output -> SGRID-annotated ``parcels_amd.Dataset`` (the producer side of the hot path).

Restates, for plain arrays (no xarray here), what ``parcels.convert.nemo_to_sgrid`` does (src/parcels/convert.py:308-408 and
its helpers :139-203): which NEMO names become ``U, V, W, time, depth, lon, lat``, which dimensions the staggered fields
live on, the SGRID padding that yields the C-grid index offsets X = 1, Y = 1, Z = 0 (tests/test_convert.py:93-104), and the
sign flip of W.  Reading the NetCDF / zarr files stays with the caller's I/O stack: pass what it returned.
"""

from __future__ import annotations

import numpy as np

from .dataset import DataArray, Dataset, _as_da
from .sgrid import FaceNodePadding, Padding, SGrid2DMetadata

_NEMO_VARNAMES = {"time_counter": "time", "depthw": "depth", "uo": "U", "vo": "V", "wo": "W"}  # convert.py:67-73
_NEMO_DEPTH_DIMS = {"depthu": "depth_center", "depthv": "depth_center", "deptht": "depth_center", "depthw": "depth"}  # :139-148
_NEMO_KEEP_DIMS = ("x", "y", "time", "x_center", "y_center", "depth", "depth_center")  # convert.py:44-56


def _field_da(name, src) -> DataArray:
    if isinstance(src, Dataset):  # convert.py:333-335: a dataset holding the field under its name
        src = src[name]
    return _as_da(src)


def nemo_to_sgrid(*, fields: dict, coords) -> Dataset:
    """fields: name -> (dims, array[, attrs]) / DataArray / Dataset containing ``name``, with NEMO dimension names
    (``time_counter, depthu|depthv|deptht|depthw, y, x``); coords: Dataset (or dict) with ``glamf, gphif`` (2-D, or with
    singleton time/depth axes) and optionally ``depthw`` and ``time_counter``/``time``.

    Returns a Dataset following the SGRID conventions that ``FieldSet.from_sgrid_conventions`` reads:
    nodes ``x, y``; faces ``x_center:x``, ``y_center:y`` (padding low), vertical ``depth_center:depth`` (padding high);
    U on ``(y_center, x)``, V on ``(y, x_center)``, everything else on the node dimensions; W negated (NEMO's W is positive
    upwards, depth increases downwards, convert.py:380-382); node coordinates renamed to ``lon, lat`` in degrees.
    """
    if not isinstance(coords, Dataset):
        coords = Dataset({}, dict(coords))
    for required in ("glamf", "gphif"):  # convert.py:37-41,126-136
        if required not in coords:
            raise ValueError(f"Expected coordinate '{required}' not found in provided coords dataset.")
    out_vars, out_coords = {}, {}

    def squeeze_2d(da: DataArray) -> DataArray:  # convert.py:348-361: drop time (length 1) and singleton axes
        a, dims = da.data, list(da.dims)
        for ax in range(a.ndim - 1, -1, -1):
            if a.ndim > 2 and a.shape[ax] == 1:
                a = np.squeeze(a, axis=ax)
                dims.pop(ax)
        if a.ndim != 2:
            if "time" in dims or "time_counter" in dims or "t" in dims:
                raise ValueError("Time dimension in coords must be length 1 (i.e., no time-varying grid).")
            raise ValueError("glamf / gphif must be 2-dimensional")
        return DataArray(("y", "x"), a, da.attrs)

    for nemo, new in (("glamf", "lon"), ("gphif", "lat")):
        da = squeeze_2d(coords[nemo])
        out_coords[new] = DataArray(da.dims, da.data, {**da.attrs, "units": "degrees"})  # convert.py:401-406

    have_depth = False
    if "depthw" in coords:
        d = coords["depthw"]
        out_coords["depth"] = DataArray(("depth",), np.asarray(d.data).reshape(-1), {**d.attrs, "axis": "Z"})
        have_depth = True
    for tname in ("time_counter", "time"):
        if tname in coords and coords[tname].data.size > 1:
            out_coords["time"] = DataArray(("time",), np.asarray(coords[tname].data).reshape(-1), {**coords[tname].attrs, "axis": "T"})

    for name, src in fields.items():
        da = _field_da(name, src)
        new_name = _NEMO_VARNAMES.get(name, name)
        dims = []
        for dname in da.dims:
            dname = _NEMO_VARNAMES.get(dname, dname)  # time_counter -> time (depthw -> depth is in the table below too)
            dname = _NEMO_DEPTH_DIMS.get(dname, dname)
            dims.append(dname)
        if new_name == "U":  # convert.py:339-343
            dims = ["y_center" if d == "y" else d for d in dims]
        elif new_name == "V":
            dims = ["x_center" if d == "x" else d for d in dims]
        from .sources import is_level_source

        if is_level_source(da.data):
            raise TypeError(f"nemo_to_sgrid needs the values of field {name!r} in memory (it drops singleton dimensions and negates W): pass "
                            "np.asarray / np.memmap here, or build the Dataset with the level source directly (dims time, depth, y, x and "
                            "W already negated: parcels_amd.Dataset, DESIGN.md section 8)")
        a = np.asarray(da.data)
        keep = [i for i, d in enumerate(dims) if d in _NEMO_KEEP_DIMS]  # convert.py:175-186: unknown dimensions are dropped
        if len(keep) != len(dims):
            if any(a.shape[i] != 1 for i in range(a.ndim) if i not in keep):
                raise ValueError(f"field {name!r} has a non-singleton dimension outside {_NEMO_KEEP_DIMS}")
            a = a.reshape([a.shape[i] for i in keep])
            dims = [dims[i] for i in keep]
        if new_name == "W":
            a = -a  # convert.py:380-382
        out_vars[new_name] = DataArray(tuple(dims), a, da.attrs)
        if any(d in ("depth", "depth_center") for d in dims):
            have_depth = have_depth or "depth" in out_coords

    if not have_depth:  # convert.py:150-154: surface data gets a single depth level 0
        out_coords["depth"] = DataArray(("depth",), np.array([0.0]), {"axis": "Z"})
        for k, da in list(out_vars.items()):
            if "depth" not in da.dims and "depth_center" not in da.dims:
                tpos = 1 if da.dims and da.dims[0] == "time" else 0
                out_vars[k] = DataArray(da.dims[:tpos] + ("depth",) + da.dims[tpos:], np.expand_dims(da.data, tpos), da.attrs)

    md = SGrid2DMetadata(  # convert.py:384-399
        cf_role="grid_topology", topology_dimension=2, node_dimensions=("x", "y"), node_coordinates=("lon", "lat"),
        face_dimensions=(FaceNodePadding("x_center", "x", Padding.LOW), FaceNodePadding("y_center", "y", Padding.LOW)),
        vertical_dimensions=(FaceNodePadding("depth_center", "depth", Padding.HIGH),),
    )
    return Dataset(out_vars, out_coords, sgrid=md)
