"""Level sources: field data that is read from disk one time level at a time.

The reference hands the hot path dask-backed arrays and its ``WindowedArray`` reads each level once, on demand
(src/parcels/_core/_windowed_array.py:56-97, _xarray.py:13-35).  The engine's counterpart of "a level is needed" is
``DeviceEngine._upload(name, level)``: with a fully materialised NumPy array (or ``np.memmap``) it slices the level; with one of the
sources below it asks the source -- for a streamed field during the previous launch (``_prefetch``), so the read, the decompression
and the page-ins overlap the RK sub-steps, and a dataset larger than the host's memory never exists in it as a whole.

A level source is anything with

    .shape  (nt, nz, ny, nx) in T, Z, Y, X order (size 1 for absent axes: put the array in TZYX order when you write it)
    .dtype  float32 or float64
    .read_level(k) -> C-contiguous ndarray of shape[1:]

``Dataset`` keeps such an object as it is (no ``np.asarray``).  NaN fill values are replaced by 0 per level at read time, like
``StructuredModelData`` does for in-memory arrays (model.py:135-143).

* ``NpyLevels``  -- a directory / list of ``.npy`` files, one per time level (memory-mapped).
* ``NetCDFLevels`` -- one variable of a NetCDF-4 (HDF5: contiguous or chunked, deflate / shuffle / fletcher32) or classic NetCDF-3
  file, or of a list of such files that continue each other in time (one file per day / month, as models write them); CF packing
  (scale_factor, add_offset, _FillValue, missing_value) is undone.  The reader is parcels_amd/hdf5.py: no netCDF4 / h5py needed.
* ``ZarrLevels`` -- one array of a zarr v2 or v3 directory store on a local filesystem, chunked with ONE time level per chunk along
  the first axis (chunks may split z / y / x), compressor null, zstd, lz4, gzip or blosc (lz4 / zstd / zlib inner codec, byte
  shuffle) -- the formats pyarrow's codecs can decode; no zarr / numcodecs installation is needed.
"""

from __future__ import annotations

import glob
import json
import os
import struct

import numpy as np

__all__ = ["LevelSource", "NetCDFLevels", "NpyLevels", "ZarrLevels", "is_level_source", "read_netcdf_variable"]


class LevelSource:
    """Base class / protocol marker."""

    shape: tuple
    dtype: np.dtype
    fill_nan = True  # replace NaN by 0 when a level is read (model.py:135-143)

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def nbytes(self):
        return int(np.prod(self.shape)) * np.dtype(self.dtype).itemsize

    def read_level(self, k: int) -> np.ndarray:  # pragma: no cover - interface
        raise NotImplementedError

    def level(self, k: int, dtype=None) -> np.ndarray:
        """Level ``k`` ready for the upload: C-contiguous, target dtype, NaN -> 0."""
        a = self.read_level(int(k))
        if tuple(a.shape) != tuple(self.shape[1:]):
            raise ValueError(f"level {k} has shape {a.shape}, expected {self.shape[1:]}")
        if self.fill_nan and np.issubdtype(a.dtype, np.floating) and np.isnan(a).any():
            a = np.nan_to_num(a, nan=0.0)
        return np.ascontiguousarray(a, dtype=dtype or self.dtype)

    def __getitem__(self, k):
        if isinstance(k, (int, np.integer)):
            return self.level(int(k))
        raise TypeError("a level source is indexed by one time level at a time")


def is_level_source(obj) -> bool:
    return isinstance(obj, LevelSource) or (hasattr(obj, "read_level") and hasattr(obj, "shape") and hasattr(obj, "dtype"))


class NpyLevels(LevelSource):
    """One ``.npy`` file per time level: ``NpyLevels("/data/U_*.npy")`` (sorted glob), a directory, or a list of paths.  Each file
    holds one (nz, ny, nx) level -- or any shape that reshapes to ``level_shape``."""

    def __init__(self, paths, level_shape=None):
        if isinstance(paths, (str, os.PathLike)):
            p = str(paths)
            paths = sorted(glob.glob(os.path.join(p, "*.npy"))) if os.path.isdir(p) else sorted(glob.glob(p))
        self.paths = [str(p) for p in paths]
        if not self.paths:
            raise ValueError("NpyLevels: no level files found")
        first = np.load(self.paths[0], mmap_mode="r")
        ls = tuple(level_shape) if level_shape is not None else tuple(first.shape)
        ls = (1,) * (3 - len(ls)) + ls if len(ls) < 3 else ls
        if int(np.prod(ls)) != int(np.prod(first.shape)) or len(ls) != 3:
            raise ValueError(f"NpyLevels: a level of shape {first.shape} does not reshape to (nz, ny, nx) = {ls}")
        self.shape = (len(self.paths),) + ls
        self.dtype = np.dtype(first.dtype)

    def read_level(self, k):
        return np.load(self.paths[k], mmap_mode="r").reshape(self.shape[1:])


def _choose_float_dtype(dtype: np.dtype, sf, ao) -> np.dtype:
    """The float dtype xarray's CF decoding gives a variable (xarray/coding/variables.py: _choose_float_dtype) -- the reference's fields
    are whatever `xr.open_dataset` decodes: the type of scale_factor / add_offset when both are given as the same float type (float64 for
    int32 data: 24 bits of mantissa do not hold it); float64 as soon as an offset comes without such a partner; the scale factor's type
    alone; without packing attributes float32 for float16 / float32 and for integers of at most 2 bytes, float64 for everything else."""
    f32, f64 = np.dtype(np.float32), np.dtype(np.float64)
    if sf is not None or ao is not None:
        st = np.asarray(sf).dtype if sf is not None else None
        ot = np.asarray(ao).dtype if ao is not None else None
        if sf is not None and ao is not None and st == ot and st in (f32, f64):
            return f64 if (dtype.kind in "iu" and dtype.itemsize == 4) else st
        if ao is not None:
            return f64
        return st if st in (f32, f64) else f64
    if dtype.kind == "f" and dtype.itemsize <= 4:
        return f32
    if dtype.kind in "iu" and dtype.itemsize <= 2:
        return f32
    return f64


def cf_unpack(a: np.ndarray, attrs: dict, fill_dtype=None) -> np.ndarray:
    """CF conventions: _FillValue / missing_value -> NaN, then packed * scale_factor + add_offset, in the float dtype xarray's decode_cf
    chooses (what the reference gets, convert.py:308-408 via xr.open_dataset; _choose_float_dtype above)."""
    sf, ao = attrs.get("scale_factor"), attrs.get("add_offset")
    fv = [np.asarray(attrs[k]).ravel() for k in ("_FillValue", "missing_value") if k in attrs]
    if a.dtype.kind == "f" and sf is None and ao is None and not fv:
        return a
    out_dt = _choose_float_dtype(a.dtype, sf, ao)
    if a.dtype.kind == "f" and a.dtype.itemsize > np.dtype(out_dt).itemsize:
        out_dt = a.dtype
    mask = None
    for v in fv:
        for x in v:
            m = np.isnan(a) if (isinstance(x, (float, np.floating)) and np.isnan(x)) else (a == x)
            mask = m if mask is None else (mask | m)
    out = a.astype(out_dt)
    if sf is not None:
        out *= np.asarray(sf, dtype=out_dt).ravel()[0]
    if ao is not None:
        out += np.asarray(ao, dtype=out_dt).ravel()[0]
    if mask is not None and mask.any():
        out[mask] = np.nan
    return out


def _open_netcdf(path):
    """HDF5File (NetCDF-4) or NetCDF3File (classic) by the magic number."""
    from . import hdf5

    with open(path, "rb") as fh:
        magic = fh.read(4)
    return hdf5.NetCDF3File(path) if magic[:3] == b"CDF" else hdf5.HDF5File(path)


def read_netcdf_variable(path, name, decode=True) -> np.ndarray:
    """A whole (small) variable -- coordinates, time axes, masks -- of a NetCDF-4 or classic NetCDF file as a NumPy array."""
    f = _open_netcdf(path)
    try:
        if hasattr(f, "vars"):
            a, attrs = f.read(name), f.vars[name]["attrs"]
        else:
            d = f.dataset(name)
            a, attrs = d.read(), d.attrs
        return cf_unpack(a, attrs) if decode else a
    finally:
        f.close()


class NetCDFLevels(LevelSource):
    """``NetCDFLevels("U_y2000.nc", "vozocrtx")`` or ``NetCDFLevels(["U_m01.nc", "U_m02.nc", ...], "vozocrtx")``: a (time, [z,] y, x)
    variable read one time level per request (for a chunked variable: only the chunks of that level are read and inflated).
    ``dims``: which of "tzyx" the variable's axes are when it has fewer than four (default "tyx" for 3-D, "yx" for 2-D)."""

    def __init__(self, paths, variable, dims=None, fill_nan=True):
        if isinstance(paths, (str, os.PathLike)):
            p = str(paths)
            paths = sorted(glob.glob(p)) if any(ch in p for ch in "*?[") else [p]
        self.paths = [str(p) for p in paths]
        if not self.paths:
            raise ValueError("NetCDFLevels: no file found")
        self.variable = variable
        self._files = [None] * len(self.paths)
        self._nt = []
        vshape = None
        for k in range(len(self.paths)):
            shp, dt, attrs = self._meta(k)
            if vshape is None:
                vshape, self._vdtype, self.attrs = shp, dt, attrs
            elif shp[1:] != vshape[1:] or dt != self._vdtype:
                raise ValueError(f"{self.paths[k]}: variable {variable!r} has shape {shp} / dtype {dt}, the first file {vshape} / {self._vdtype}")
            self._nt.append(shp[0] if self._has_t(shp, dims) else 1)
        nd = len(vshape)
        dims = dims or {4: "tzyx", 3: "tyx", 2: "yx", 1: "x"}.get(nd)
        if dims is None or len(dims) != nd or any(c not in "tzyx" for c in dims) or list(dims) != sorted(dims, key="tzyx".index):
            raise ValueError(f"NetCDFLevels: variable {variable!r} has {nd} axes; dims={dims!r} must name them in t, z, y, x order")
        self._dims = dims
        self._offsets = np.concatenate([[0], np.cumsum(self._nt)])
        level = tuple(vshape[dims.index(c)] if c in dims else 1 for c in "zyx")
        self.shape = (int(self._offsets[-1]),) + level
        probe = cf_unpack(np.zeros(1, self._vdtype), self.attrs)
        self.dtype = np.dtype(np.float32 if probe.dtype == np.float32 else np.float64)
        self.fill_nan = fill_nan

    @staticmethod
    def _has_t(shp, dims):
        return (dims or {4: "tzyx", 3: "tyx"}.get(len(shp), "")).startswith("t")

    def _file(self, k):
        if self._files[k] is None:
            self._files[k] = _open_netcdf(self.paths[k])
        return self._files[k]

    def _meta(self, k):
        f = self._file(k)
        if hasattr(f, "vars"):
            v = f.vars[self.variable]
            return tuple(f.shape(self.variable)), v["dtype"].newbyteorder("="), v["attrs"]
        d = f.dataset(self.variable)
        return tuple(d.shape), np.dtype(d.dtype).newbyteorder("="), d.attrs

    def read_level(self, k):
        fi = int(np.searchsorted(self._offsets, k, side="right") - 1)
        local = int(k - self._offsets[fi])
        f = self._file(fi)
        first = local if self._dims.startswith("t") else None
        a = f.read(self.variable, first) if hasattr(f, "vars") else f.dataset(self.variable).read(first)
        return cf_unpack(a, self.attrs).reshape(self.shape[1:])

    def close(self):
        for k, f in enumerate(self._files):
            if f is not None:
                f.close()
                self._files[k] = None


# ---- zarr v2 ----------------------------------------------------------------------------------------------------------------
def _blosc_decode(buf: bytes) -> bytes:
    """One Blosc-1 frame (what numcodecs.Blosc writes): 16-byte header, block offsets, per-block [split] streams."""
    import pyarrow as pa

    version, versionlz, flags, typesize = struct.unpack_from("<BBBB", buf, 0)
    nbytes, blocksize, cbytes = struct.unpack_from("<III", buf, 4)
    if version != 2:
        raise ValueError(f"unsupported blosc frame version {version}")
    if flags & 0x2:  # memcpyed: the payload follows the header verbatim
        return bytes(buf[16:16 + nbytes])
    if flags & 0x4:
        raise ValueError("blosc bit-shuffle is not supported (use byte shuffle or no shuffle)")
    codec = {0: None, 1: "lz4", 2: "lz4", 4: "gzip", 5: "zstd"}.get(flags >> 5, "?")
    if codec in (None, "?"):
        raise ValueError(f"unsupported blosc inner codec {flags >> 5} (blosclz / snappy): re-encode with lz4, zstd or zlib")
    shuffle = bool(flags & 0x1) and typesize > 1
    dont_split = bool(flags & 0x10)
    nblocks = (nbytes + blocksize - 1) // blocksize
    offs = struct.unpack_from(f"<{nblocks}I", buf, 16)
    out = bytearray(nbytes)
    cod = pa.Codec("lz4_raw" if codec == "lz4" else codec)
    for b in range(nblocks):
        bsize = min(blocksize, nbytes - b * blocksize)
        # blosc splits a full block into `typesize` byte-plane streams unless told not to; the short last block is never split
        split = (not dont_split) and bsize == blocksize and 1 < typesize <= 16 and blocksize // typesize >= 128
        nsplit = typesize if split else 1
        pos = offs[b]
        block = bytearray()
        for _ in range(nsplit):
            (cb,) = struct.unpack_from("<i", buf, pos)
            pos += 4
            neblock = bsize // nsplit
            chunk = buf[pos:pos + cb]
            block += chunk if cb == neblock else cod.decompress(chunk, decompressed_size=neblock).to_pybytes()  # cb == raw size: stored
            pos += cb
        if shuffle:
            n = bsize // typesize
            arr = np.frombuffer(bytes(block[: n * typesize]), dtype=np.uint8).reshape(typesize, n).T.reshape(-1)
            block = bytearray(arr.tobytes()) + block[n * typesize:]
        out[b * blocksize:b * blocksize + bsize] = block
    return bytes(out)


class ZarrLevels(LevelSource):
    """``ZarrLevels("/data/model.zarr", "uo")``: the array ``uo`` of a zarr directory store whose first axis is time with chunk
    length 1 (a (time, y, x) surface field gets a depth axis of size 1).

    * zarr **v2** (``.zarray``): C order, no filters; compressor null, zstd, lz4, zlib / gzip or blosc.
    * zarr **v3** (``zarr.json``): regular chunk grid, ``default`` ("c/0/1/2") or ``v2`` chunk keys, codec chain ``bytes`` (little
      endian) optionally followed by ONE of ``zstd``, ``gzip``, ``blosc``; no sharding, no transpose."""

    def __init__(self, store, array, fill_nan=True):
        self.root = os.path.join(str(store), array)
        v2, v3 = os.path.join(self.root, ".zarray"), os.path.join(self.root, "zarr.json")
        if os.path.exists(v2):
            meta = json.load(open(v2))
            if meta.get("zarr_format") != 2 or meta.get("order", "C") != "C" or meta.get("filters"):
                raise ValueError("ZarrLevels reads zarr v2 arrays in C order without filters")
            self.zshape = tuple(meta["shape"])
            self.chunks = tuple(meta["chunks"])
            self.zdtype = np.dtype(meta["dtype"])
            self.compressor = meta.get("compressor")
            self.fill_value = meta.get("fill_value")
            sep = meta.get("dimension_separator", ".")
            self._key = lambda idx: sep.join(str(v) for v in idx)
        elif os.path.exists(v3):
            meta = json.load(open(v3))
            if meta.get("zarr_format") != 3 or meta.get("node_type") != "array":
                raise ValueError(f"{v3}: not a zarr v3 array")
            grid = meta.get("chunk_grid", {})
            if grid.get("name") != "regular":
                raise ValueError("ZarrLevels reads zarr v3 arrays with a regular chunk grid")
            self.zshape = tuple(meta["shape"])
            self.chunks = tuple(grid["configuration"]["chunk_shape"])
            self.zdtype = np.dtype({"float32": "<f4", "float64": "<f8", "int16": "<i2", "int32": "<i4", "int64": "<i8"}[meta["data_type"]])
            self.fill_value = meta.get("fill_value")
            cke = meta.get("chunk_key_encoding", {"name": "default"})
            sep = (cke.get("configuration") or {}).get("separator", "/" if cke.get("name", "default") == "default" else ".")
            self._key = (lambda idx: "c" + sep + sep.join(str(v) for v in idx)) if cke.get("name", "default") == "default" else (lambda idx: sep.join(str(v) for v in idx))
            self.compressor = None
            for c in meta.get("codecs", []):
                name, conf = c.get("name"), c.get("configuration") or {}
                if name == "bytes":
                    if conf.get("endian", "little") != "little":
                        raise ValueError("ZarrLevels reads little-endian zarr v3 arrays")
                elif name in ("zstd", "gzip", "blosc") and self.compressor is None:
                    self.compressor = {"id": name}
                else:
                    raise ValueError(f"unsupported zarr v3 codec chain (codec {name!r}): bytes [+ zstd | gzip | blosc] is read")
        else:
            raise FileNotFoundError(f"{self.root}: neither .zarray (zarr v2) nor zarr.json (zarr v3)")
        if self.chunks[0] != 1:
            raise ValueError(f"the time axis must be chunked one level per chunk, got chunks={self.chunks}")
        # CF packing attributes (.zattrs of zarr v2, "attributes" of zarr.json): packed integers are unpacked level by level
        self.attrs = {}
        za = os.path.join(self.root, ".zattrs")
        raw_attrs = json.load(open(za)) if os.path.exists(za) else (meta.get("attributes") or {})
        for key in ("scale_factor", "add_offset", "_FillValue", "missing_value"):
            if isinstance(raw_attrs.get(key), (int, float)):
                self.attrs[key] = np.float32(raw_attrs[key]) if (key in ("scale_factor", "add_offset") and self.zdtype.itemsize <= 2) else (
                    np.float64(raw_attrs[key]) if key in ("scale_factor", "add_offset") else np.asarray(raw_attrs[key]).astype(self.zdtype if self.zdtype.kind != "f" else np.float64))
        self.dtype = np.dtype(cf_unpack(np.zeros(1, self.zdtype), self.attrs).dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            self.dtype = np.dtype(np.float64)
        rest = self.zshape[1:]
        if not (1 <= len(rest) <= 3):
            raise ValueError(f"expected a (time, [z,] [y,] x) array, got shape {self.zshape}")
        self.shape = (self.zshape[0],) + (1,) * (3 - len(rest)) + tuple(rest)
        self.fill_nan = fill_nan

    def _decode(self, raw: bytes) -> bytes:
        c = self.compressor
        if c is None:
            return raw
        cid = c.get("id")
        if cid == "blosc":
            return _blosc_decode(raw)
        import pyarrow as pa

        if cid == "zstd":
            n = int(np.prod(self.chunks)) * self.zdtype.itemsize
            return pa.Codec("zstd").decompress(raw, decompressed_size=n).to_pybytes()
        if cid == "lz4":  # numcodecs.LZ4: 4-byte little-endian size + raw lz4 block
            (n,) = struct.unpack_from("<I", raw, 0)
            return pa.Codec("lz4_raw").decompress(raw[4:], decompressed_size=n).to_pybytes()
        if cid in ("zlib", "gzip"):
            import zlib

            return zlib.decompress(raw) if cid == "zlib" else zlib.decompress(raw, 31)
        raise ValueError(f"unsupported zarr compressor {cid!r}")

    def read_level(self, k):
        rest_shape, rest_chunks = self.zshape[1:], self.chunks[1:]
        out = np.empty(rest_shape, dtype=self.zdtype)
        missing = None  # cells of unwritten chunks of an INTEGER array whose fill value is null / "NaN": NaN after unpacking
        grid = [range((s + c - 1) // c) for s, c in zip(rest_shape, rest_chunks)]
        for idx in np.ndindex(*[len(g) for g in grid]):
            path = os.path.join(self.root, self._key((k,) + tuple(idx)))
            sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, rest_chunks, rest_shape))
            if not os.path.exists(path):  # an unwritten chunk holds the fill value
                nanfill = self.fill_value in (None, "NaN")  # "NaN": the JSON spelling of both formats
                if nanfill and self.zdtype.kind != "f":
                    out[sl] = 0
                    if missing is None:
                        missing = np.zeros(rest_shape, dtype=bool)
                    missing[sl] = True
                else:
                    out[sl] = np.nan if nanfill else self.fill_value
                continue
            buf = self._decode(open(path, "rb").read())
            chunk = np.frombuffer(buf, dtype=self.zdtype, count=int(np.prod(rest_chunks))).reshape(rest_chunks)
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
        out = cf_unpack(out, self.attrs)
        if missing is not None:
            out = out.astype(self.dtype, copy=False)
            out[missing] = np.nan
        return out.reshape(self.shape[1:])
