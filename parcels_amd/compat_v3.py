"""Particle file (Parquet, one row per particle and output time) -> Parcels-v3 style ``(trajectory, obs)`` zarr store
(the reference's ``particlefile_to_v3_zarr``, src/parcels/_compat_v3.py:30-110; SURVEY.md section 8(f) item 1, optional tail).

The reference pivots with polars and writes through xarray; neither is a dependency here.  The pivot is two NumPy passes (a
lexicographic sort by (trajectory, time) and a scatter into dense ``(trajectory, obs)`` arrays, missing observations NaN), the store a
zarr **format 2** directory written directly -- ``.zgroup``, one ``.zarray`` / ``.zattrs`` pair and zlib-compressed chunks per
variable, xarray's ``_ARRAY_DIMENSIONS`` attribute and consolidated ``.zmetadata`` -- so that ``xarray.open_zarr`` reads it as the
v3 layout (dims ``trajectory``, ``obs``, both also coordinates; variables ``lon``, ``lat``, ``z``, ``time`` + whatever else was
written; field and file metadata carried over), as tests/test_compat_v3.py of the reference asks.
"""

from __future__ import annotations

import json
import os
import zlib
from pathlib import Path

import numpy as np

__all__ = ["particlefile_to_v3_zarr"]

_RENAME = {"particle_id": "trajectory", "t": "time", "x": "lon", "y": "lat"}  # _compat_v3.py:64


def _utf8(d) -> dict:
    out = {}
    for k, v in (d or {}).items():
        k = k.decode("utf8") if isinstance(k, bytes) else k
        if isinstance(v, dict):
            v = _utf8(v)
        elif isinstance(v, bytes):
            v = v.decode("utf8")
        out[k] = v
    return out


def _zarr_dtype(dt: np.dtype) -> str:
    dt = np.dtype(dt)
    if dt.kind == "b":
        return "|b1"
    if dt.itemsize == 1:
        return "|" + dt.kind + "1"
    return "<" + dt.kind + str(dt.itemsize)


def _write_array(root: Path, name: str, a: np.ndarray, dims, attrs: dict, chunk_rows=4096):
    """One zarr v2 array: C order, chunks of up to ``chunk_rows`` rows (all of the other axis), zlib level 1."""
    a = np.ascontiguousarray(a)
    if a.dtype.byteorder == ">":
        a = a.astype(a.dtype.newbyteorder("<"))
    d = root / name
    d.mkdir(parents=True, exist_ok=True)
    chunks = [max(1, min(chunk_rows, a.shape[0]))] + [max(1, s) for s in a.shape[1:]]
    if a.dtype.kind == "f":
        fill = "NaN"
    elif a.dtype.kind == "b":
        fill = False
    else:
        fill = 0
    meta = {"zarr_format": 2, "shape": list(a.shape), "chunks": chunks, "dtype": _zarr_dtype(a.dtype), "compressor": {"id": "zlib", "level": 1},
            "fill_value": fill, "order": "C", "filters": None}
    zattrs = dict(attrs)
    zattrs["_ARRAY_DIMENSIONS"] = list(dims)
    (d / ".zarray").write_text(json.dumps(meta, indent=1))
    (d / ".zattrs").write_text(json.dumps(zattrs, indent=1))
    if a.size:
        for ci, lo in enumerate(range(0, a.shape[0], chunks[0])):
            block = a[lo:lo + chunks[0]]
            if block.shape[0] != chunks[0]:  # zarr chunks are always full-sized: pad the last one with the fill value
                pad = np.full([chunks[0]] + list(a.shape[1:]), np.nan if a.dtype.kind == "f" else 0, dtype=a.dtype)
                pad[: block.shape[0]] = block
                block = pad
            key = ".".join([str(ci)] + ["0"] * (a.ndim - 1))
            (d / key).write_bytes(zlib.compress(block.tobytes(), 1))
    return {name + "/.zarray": meta, name + "/.zattrs": zattrs}


def particlefile_to_v3_zarr(from_parquet, to_zarr) -> None:
    """Convert a particle file (path or file-like object holding Parquet) to a v3-style zarr store at ``to_zarr`` (must end in
    ``.zarr``).  ``particle_id -> trajectory``, ``t -> time``, ``x -> lon``, ``y -> lat``; observation k of a trajectory is its k-th row
    in time order.  Not lazy: the whole file is pivoted in memory, like the reference's."""
    import pyarrow.parquet as pq

    to_zarr = Path(to_zarr)
    if to_zarr.suffix != ".zarr":
        raise ValueError(f"Parameter `to_zarr` must have a '.zarr' suffix. Got {to_zarr=}.")
    table = pq.read_table(from_parquet)
    names = list(table.schema.names)
    missing = [k for k in _RENAME if k not in names]
    if missing:
        raise KeyError(f"Expected to have all columns {list(_RENAME)} in the output parquet. Got columns {names}.")
    cols, field_meta = {}, {}
    for nm in names:
        z = _RENAME.get(nm, nm)
        col = table.column(nm)
        a = col.to_numpy() if col.null_count == 0 else np.asarray(col.to_pandas())
        if a.dtype.kind in "mM":  # temporal columns: the integer count + its unit (what xarray decodes)
            unit = np.datetime_data(a.dtype)[0]
            field_meta.setdefault(z, {})["parquet_unit"] = unit
            a = a.astype("int64")
        cols[z] = a
        field_meta[z] = {**field_meta.get(z, {}), **_utf8(table.field(nm).metadata)}

    traj, tm = cols["trajectory"], cols["time"]
    order = np.lexsort((tm, traj))  # by trajectory, then time (stable: equal times keep file order)
    st = traj[order]
    ids, first, counts = np.unique(st, return_index=True, return_counts=True)
    n = len(st)
    row = np.repeat(np.arange(len(ids)), counts)          # trajectory index of every sorted row
    obs = np.arange(n) - np.repeat(first, counts)        # its observation number
    nobs = int(counts.max()) if n else 0
    complete = n == len(ids) * nobs

    if to_zarr.exists():
        raise FileExistsError(f"{to_zarr} exists")
    to_zarr.mkdir(parents=True)
    file_attrs = _utf8(table.schema.metadata)
    file_attrs.pop("ARROW:schema", None)
    file_attrs.pop("pandas", None)
    (to_zarr / ".zgroup").write_text(json.dumps({"zarr_format": 2}))
    (to_zarr / ".zattrs").write_text(json.dumps(file_attrs, indent=1))
    consolidated = {".zgroup": {"zarr_format": 2}, ".zattrs": file_attrs}
    consolidated.update(_write_array(to_zarr, "trajectory", ids, ["trajectory"], field_meta.get("trajectory", {})))
    consolidated.update(_write_array(to_zarr, "obs", np.arange(nobs, dtype=np.int64), ["obs"], {}))
    for z, a in cols.items():
        if z == "trajectory":
            continue
        a = a[order]
        if complete:
            dense = a.reshape(len(ids), nobs)
        else:  # ragged trajectories: missing observations are NaN (integers become float64, as a pivot with nulls does)
            dt = a.dtype if a.dtype.kind == "f" else np.dtype("float64")
            dense = np.full((len(ids), nobs), np.nan, dtype=dt)
            dense[row, obs] = a
        consolidated.update(_write_array(to_zarr, z, dense, ["trajectory", "obs"], field_meta.get(z, {})))
    (to_zarr / ".zmetadata").write_text(json.dumps({"zarr_consolidated_format": 1, "metadata": consolidated}, indent=1))
    if not os.path.isdir(to_zarr):  # pragma: no cover
        raise OSError(f"could not create {to_zarr}")
