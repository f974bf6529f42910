"""Level sources (parcels_amd/sources.py): field data read from disk one time level at a time -- per-level .npy files and zarr v2
arrays (null / zstd / lz4 / blosc chunks), the producer side of the device's level ring."""

from __future__ import annotations

import json
import os
import struct

import numpy as np
import pytest

import parcels_amd as pa
from case_utils import build_fieldset, build_pset, compare


def _write_zarr_v2(root, name, arr, chunks, compressor):
    """A zarr v2 directory array written by hand (zarr / numcodecs are not installed): null, zstd, numcodecs-style lz4."""
    import pyarrow as pyarrow

    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    meta = {"zarr_format": 2, "shape": list(arr.shape), "chunks": list(chunks), "dtype": arr.dtype.str, "order": "C", "filters": None,
            "fill_value": "NaN", "compressor": None if compressor is None else {"id": compressor}}
    json.dump(meta, open(os.path.join(d, ".zarray"), "w"))
    grid = [range((s + c - 1) // c) for s, c in zip(arr.shape, chunks)]
    for idx in np.ndindex(*[len(g) for g in grid]):
        chunk = np.full(chunks, np.nan, arr.dtype)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, arr.shape))
        part = arr[sl]
        chunk[tuple(slice(0, n) for n in part.shape)] = part
        raw = chunk.tobytes()
        if compressor == "zstd":
            raw = pyarrow.Codec("zstd").compress(raw).to_pybytes()
        elif compressor == "lz4":
            raw = struct.pack("<I", len(raw)) + pyarrow.Codec("lz4_raw").compress(raw).to_pybytes()
        open(os.path.join(d, ".".join(str(i) for i in idx)), "wb").write(raw)


@pytest.mark.parametrize("compressor", [None, "zstd", "lz4"])
def test_zarr_levels_reads_what_was_written(tmp_path, compressor):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((5, 4, 13, 17)).astype(np.float32)
    a[2, 1, 3, 4] = np.nan
    _write_zarr_v2(str(tmp_path), "U", a, (1, 2, 8, 17), compressor)
    src = pa.ZarrLevels(str(tmp_path), "U")
    assert src.shape == a.shape and src.dtype == np.float32
    for k in range(5):
        assert np.array_equal(src.read_level(k), a[k], equal_nan=True)
    assert src.level(2)[1, 3, 4] == 0.0  # NaN fill -> 0 like model.py:135-143
    b = rng.standard_normal((3, 9, 11))  # (time, y, x) surface field -> (nt, 1, ny, nx)
    _write_zarr_v2(str(tmp_path), "S", b, (1, 9, 11), compressor)
    s2 = pa.ZarrLevels(str(tmp_path), "S")
    assert s2.shape == (3, 1, 9, 11) and np.array_equal(s2.level(1)[0], b[1])
    with pytest.raises(ValueError):
        _write_zarr_v2(str(tmp_path), "bad", a, (2, 4, 13, 17), None)
        pa.ZarrLevels(str(tmp_path), "bad")


def _write_zarr_v3(root, name, arr, chunks, codec, key_encoding="default"):
    """A zarr v3 directory array written by hand: zarr.json, "c/<i>/<j>/..." (or v2-style "i.j.k") chunk keys, codecs bytes [+ zstd | gzip]."""
    import zlib

    import pyarrow as pyarrow

    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    codecs = [{"name": "bytes", "configuration": {"endian": "little"}}]
    if codec:
        codecs.append({"name": codec, "configuration": {"level": 1}})
    meta = {"zarr_format": 3, "node_type": "array", "shape": list(arr.shape), "data_type": {"<f4": "float32", "<f8": "float64"}[arr.dtype.str],
            "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": list(chunks)}},
            "chunk_key_encoding": {"name": key_encoding, "configuration": {"separator": "/" if key_encoding == "default" else "."}},
            "fill_value": "NaN", "codecs": codecs, "attributes": {}}
    json.dump(meta, open(os.path.join(d, "zarr.json"), "w"))
    grid = [range((s + c - 1) // c) for s, c in zip(arr.shape, chunks)]
    for idx in np.ndindex(*[len(g) for g in grid]):
        chunk = np.full(chunks, np.nan, arr.dtype)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, arr.shape))
        part = arr[sl]
        chunk[tuple(slice(0, n) for n in part.shape)] = part
        raw = chunk.tobytes()
        if codec == "zstd":
            raw = pyarrow.Codec("zstd").compress(raw).to_pybytes()
        elif codec == "gzip":
            co = zlib.compressobj(1, zlib.DEFLATED, 31)
            raw = co.compress(raw) + co.flush()
        key = os.path.join("c", *[str(i) for i in idx]) if key_encoding == "default" else ".".join(str(i) for i in idx)
        path = os.path.join(d, key)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        open(path, "wb").write(raw)


@pytest.mark.parametrize("codec, key_encoding", [(None, "default"), ("zstd", "default"), ("gzip", "default"), ("zstd", "v2")])
def test_zarr_v3_levels_read_what_was_written(tmp_path, codec, key_encoding):
    rng = np.random.default_rng(2)
    a = rng.standard_normal((4, 3, 10, 21)).astype(np.float32)
    a[1, 2, 3, 4] = np.nan
    _write_zarr_v3(str(tmp_path), "V", a, (1, 2, 6, 21), codec, key_encoding)
    src = pa.ZarrLevels(str(tmp_path), "V")
    assert src.shape == a.shape and src.dtype == np.float32
    for k in range(4):
        assert np.array_equal(src.read_level(k), a[k], equal_nan=True)
    assert src.level(1)[2, 3, 4] == 0.0
    os.remove(os.path.join(str(tmp_path), "V", "c", "3", "1", "1", "0") if key_encoding == "default" else os.path.join(str(tmp_path), "V", "3.1.1.0"))
    missing = src.read_level(3)  # an unwritten chunk holds the fill value
    assert np.isnan(missing[2:3, 6:10]).all() and np.array_equal(missing[:2], a[3, :2])
    b = rng.standard_normal((3, 7, 9))
    _write_zarr_v3(str(tmp_path), "S", b, (1, 7, 9), codec, key_encoding)
    assert pa.ZarrLevels(str(tmp_path), "S").shape == (3, 1, 7, 9)
    meta = json.load(open(os.path.join(str(tmp_path), "V", "zarr.json")))
    meta["codecs"].insert(0, {"name": "transpose", "configuration": {"order": [0, 1, 3, 2]}})
    json.dump(meta, open(os.path.join(str(tmp_path), "V", "zarr.json"), "w"))
    with pytest.raises(ValueError, match="codec chain"):
        pa.ZarrLevels(str(tmp_path), "V")


def test_zarr_levels_decodes_the_reference_s_blosc_stores():
    """The zarr stores the reference ships (tests/test_data/*.zarr, numcodecs Blosc lz4 + byte shuffle) through the product's own
    frame decoder, against oracle/mini_zarr.py (build container only)."""
    root = "/root/reference/tests/test_data/test_interpolation_jit_linear.zarr"
    if not os.path.isdir(root):
        pytest.skip("reference test data not present")
    from oracle import mini_zarr
    from parcels_amd.sources import _blosc_decode

    n = 0
    for name in sorted(os.listdir(root)):
        d = os.path.join(root, name)
        if not os.path.exists(os.path.join(d, ".zarray")):
            continue
        meta = json.load(open(os.path.join(d, ".zarray")))
        if (meta.get("compressor") or {}).get("id") != "blosc":
            continue
        for f in os.listdir(d):
            if f.startswith("."):
                continue
            raw = open(os.path.join(d, f), "rb").read()
            assert _blosc_decode(raw) == mini_zarr.blosc_decompress(raw)
            n += 1
    assert n > 0


def test_npy_levels(tmp_path):
    a = np.arange(4 * 3 * 5 * 6, dtype=np.float64).reshape(4, 3, 5, 6)
    for k in range(4):
        np.save(tmp_path / f"U_{k:03d}.npy", a[k])
    src = pa.NpyLevels(str(tmp_path / "U_*.npy"))
    assert src.shape == a.shape and src.dtype == np.float64
    assert np.array_equal(src[2], a[2]) and src[2].flags["C_CONTIGUOUS"]
    with pytest.raises(TypeError):
        src[1:3]
    with pytest.raises(ValueError):
        pa.NpyLevels(str(tmp_path / "nothing_*.npy"))


def _case_on_disk(tmp_path, kind):
    from oracle import cases

    case = cases.rect_agrid_case("src_" + kind, mesh="spherical", kernels=["AdvectionRK4_3D"], seed=17, nx=30, ny=20, nz=6, nt=7, npart=3000, with_w=True,
                                 field_dtype=np.float32, level_dt=43200.0, dt=3600.0, runtime=2.9 * 86400.0)
    lazy = dict(case)
    lazy["fields"] = {}
    for name, arr in case["fields"].items():
        if kind == "npy":
            d = tmp_path / name
            d.mkdir()
            for k in range(arr.shape[0]):
                np.save(d / f"{k:04d}.npy", arr[k])
            lazy["fields"][name] = pa.NpyLevels(str(d))
        elif kind == "netcdf3":  # classic NetCDF, two files that continue each other in time (what scipy can write without libhdf5)
            from scipy.io import netcdf_file

            paths = []
            for part, sl in enumerate((slice(0, 3), slice(3, None))):
                p = str(tmp_path / f"{name}_{part}.nc")
                with netcdf_file(p, "w", version=2) as nc:
                    nc.createDimension("time_counter", None)
                    for dn, n in zip(("z", "y", "x"), arr.shape[1:]):
                        nc.createDimension(dn, n)
                    v = nc.createVariable(name, "f4", ("time_counter", "z", "y", "x"))
                    for k, lvl in enumerate(arr[sl]):
                        v[k] = lvl
                paths.append(p)
            lazy["fields"][name] = pa.NetCDFLevels(paths, name)
        else:
            _write_zarr_v2(str(tmp_path / "store.zarr"), name, arr, (1, 3, 20, 16), "zstd")
            lazy["fields"][name] = pa.ZarrLevels(str(tmp_path / "store.zarr"), name)
    return case, lazy


def test_fieldset_accepts_level_sources(tmp_path):
    case, lazy = _case_on_disk(tmp_path, "npy")
    fs = build_fieldset(lazy)
    assert isinstance(fs.U.data.data, pa.NpyLevels) and fs.U.data.shape == case["fields"]["U"].shape
    assert fs.time_interval is not None and "UVW" in fs.fields


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["npy", "zarr", "netcdf3"])
def test_advection_from_level_sources_equals_in_memory_fields(gpu, tmp_path, kind):
    """Levels read from disk on demand into a ring of 3 (and all at once into a resident device copy) give the trajectories of the
    in-memory NumPy fields, bit for bit."""
    case, lazy = _case_on_disk(tmp_path, kind)
    ref_fs = build_fieldset(case)
    ref = build_pset(case, ref_fs)
    ref.execute([pa.AdvectionRK4_3D, pa.DeleteParticle], dt=case["dt"], runtime=case["runtime"])
    want = {k: np.array(v) for k, v in ref._data.items()}
    for ns in (3, None):
        fs = build_fieldset(lazy)
        fs.to_device(nslots=ns)
        pset = build_pset(lazy, fs)
        pset.execute([pa.AdvectionRK4_3D, pa.DeleteParticle], dt=case["dt"], runtime=case["runtime"])
        if ns is not None:
            assert pset._last_stats["launches"] > 1
        compare({k: np.array(v) for k, v in pset._data.items()}, want, rtol=0.0, check_state="all", label=f"{kind} nslots={ns}", skip=())


class _CountingLevels(pa.LevelSource):
    """A level source that counts how often each level is read (the `loads` counter of the reference's WindowedArray)."""

    def __init__(self, inner):
        self.inner, self.shape, self.dtype = inner, inner.shape, inner.dtype
        self.reads = {}

    def read_level(self, k):
        self.reads[k] = self.reads.get(k, 0) + 1
        return self.inner.read_level(k)


@pytest.mark.gpu
@pytest.mark.parametrize("direction", [1, -1])
def test_streamed_levels_are_read_once_and_at_most_three_are_resident(gpu, tmp_path, direction):
    """tests/test_windowed_array.py:15-80 of the reference (each time level read exactly once, forward and backward, only the
    bracketing levels resident), for the device ring: every level the run touches is read from its source exactly once, levels the
    run never reaches are never read, and the ring never holds more than its three slots."""
    case, lazy = _case_on_disk(tmp_path, "npy")
    for name in list(lazy["fields"]):
        lazy["fields"][name] = _CountingLevels(lazy["fields"][name])
    fs = build_fieldset(lazy)
    fs.to_device(nslots=3)
    pset = build_pset(lazy, fs)
    dt = case["dt"] * direction
    if direction < 0:
        pset._data["t"][:] = float(case["time_s"][-1])
    resident_max = 0
    eng = fs._engine_or_create()
    upload = eng._upload

    def counting_upload(*a, **k):
        nonlocal resident_max
        r = upload(*a, **k)
        resident_max = max(resident_max, max(len([l for l in eng._slots(n) if l >= 0]) for n in ("U", "V", "W")))
        return r

    eng._upload = counting_upload
    pset.execute([pa.AdvectionRK4_3D, pa.DeleteParticle], dt=dt, runtime=case["runtime"])
    nt = len(case["time_s"])
    span = case["runtime"] / (case["time_s"][1] - case["time_s"][0])  # levels the clock crosses
    for name, src in lazy["fields"].items():
        assert all(v == 1 for v in src.reads.values()), (name, src.reads)  # never twice
        touched = sorted(src.reads)
        assert len(touched) >= int(span) + 1 and len(touched) <= int(span) + 3, (name, touched)
        assert (touched[0] == 0) if direction > 0 else (touched[-1] == nt - 1)
        assert touched == list(range(touched[0], touched[-1] + 1))  # contiguous: nothing skipped, nothing beyond the run read
    assert 2 <= resident_max <= 3


@pytest.mark.gpu
def test_advection_from_a_netcdf4_file_equals_in_memory_fields(gpu):
    """The reference's own NetCDF-4 input (tests/test_data/test_interpolation_data_random_linear.nc, re-packed by h5repack to one
    deflated chunk per time level: tests/golden/hdf5/) as U, V, W of a FieldSet: read one level per request into a ring of 3 device
    slots -- every level read exactly once -- and all at once; the trajectories equal those of the in-memory arrays, bit for bit."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hdf5", "reference_linear_chunked_gzip.nc")
    rd = lambda n: pa.read_netcdf_variable(path, n)
    lon, lat, depth, time_s = rd("lon"), rd("lat"), rd("depth"), rd("time")
    rng = np.random.default_rng(3)
    n = 2000
    case = dict(name="nc4", mesh="flat", lon=lon, lat=lat, depth=depth, x_pad="low", y_pad="low", z_pad="both", time_s=time_s - time_s[0],
                fields={k: rd(k) for k in "UVW"}, field_dims={k: ("time", "depth", "YG", "XG") for k in "UVW"}, cgrid=False,
                kernels=["AdvectionRK4_3D", "DeleteParticle"], spatial_dtype="float64",
                x=rng.uniform(lon[2], lon[-3], n), y=rng.uniform(lat[2], lat[-3], n), z=rng.uniform(depth[1], depth[-2], n), t0=None,
                dt=float(time_s[1] - time_s[0]) / 4, runtime=float(time_s[-1] - time_s[0]) * 0.8, seed=0)
    ref = build_pset(case, build_fieldset(case))
    ref.execute([pa.AdvectionRK4_3D, pa.DeleteParticle], dt=case["dt"], runtime=case["runtime"])
    want = {k: np.array(v) for k, v in ref._data.items()}
    assert len(want["x"]) > 0 and ref._last_stats["steps"] > 10 * len(want["x"])
    for ns in (3, None):
        lazy = dict(case)
        lazy["fields"] = {k: _CountingLevels(pa.NetCDFLevels(path, k)) for k in "UVW"}
        fs = build_fieldset(lazy)
        fs.to_device(nslots=ns)
        pset = build_pset(lazy, fs)
        pset.execute([pa.AdvectionRK4_3D, pa.DeleteParticle], dt=case["dt"], runtime=case["runtime"])
        compare({k: np.array(v) for k, v in pset._data.items()}, want, rtol=0.0, check_state="all", label=f"netcdf4 nslots={ns}", skip=())
        for src in lazy["fields"].values():
            assert all(v == 1 for v in src.reads.values()), src.reads
            if ns is not None:
                assert pset._last_stats["launches"] > 1 and len(src.reads) < len(time_s)  # levels beyond the run are never read


def test_zarr_levels_unpack_cf_packed_integers(tmp_path):
    """An int16 zarr array with scale_factor / add_offset in .zattrs and an unwritten chunk (fill_value null): values are unpacked
    per level, the missing chunk reads as NaN (then 0), nothing raises (`cannot convert float NaN to integer`)."""
    rng = np.random.default_rng(5)
    raw = rng.integers(-3000, 3000, (3, 2, 8, 10)).astype(np.int16)
    _write_zarr_v2(str(tmp_path), "P", raw.astype(np.float32), (1, 2, 4, 10), None)  # writes float chunks: rewrite them as int16 below
    d = os.path.join(str(tmp_path), "P")
    meta = json.load(open(os.path.join(d, ".zarray")))
    meta["dtype"], meta["fill_value"] = "<i2", None
    json.dump(meta, open(os.path.join(d, ".zarray"), "w"))
    for k in range(3):
        for j in range(2):
            open(os.path.join(d, f"{k}.0.{j}.0"), "wb").write(np.ascontiguousarray(raw[k, :, 4 * j:4 * j + 4, :]).tobytes())
    os.remove(os.path.join(d, "1.0.1.0"))
    json.dump({"scale_factor": 0.001, "add_offset": 2.0}, open(os.path.join(d, ".zattrs"), "w"))
    src = pa.ZarrLevels(str(tmp_path), "P")
    assert src.dtype == np.float32
    lv = src.read_level(1)
    assert np.isnan(lv[:, 4:, :]).all() and np.allclose(lv[:, :4, :], raw[1, :, :4, :].astype(np.float32) * np.float32(0.001) + np.float32(2.0), rtol=1e-6)
    assert np.all(src.level(1)[:, 4:, :] == 0.0)


def test_cf_unpack_chooses_the_float_dtype_xarray_chooses():
    """xarray/coding/variables.py::_choose_float_dtype decides the dtype of a decoded field in the reference (xr.open_dataset): the
    device field precision must be the same."""
    from parcels_amd.sources import cf_unpack

    f32, f64 = np.float32, np.float64
    i16, i32, i8 = np.arange(4, dtype=np.int16), np.arange(4, dtype=np.int32), np.arange(4, dtype=np.int8)
    # both attributes, same float type: that type -- except for int32 data (24 bits of mantissa do not hold it)
    assert cf_unpack(i16, {"scale_factor": f32(0.5), "add_offset": f32(1.0)}).dtype == f32
    assert cf_unpack(i32, {"scale_factor": f32(0.5), "add_offset": f32(1.0)}).dtype == f64
    assert cf_unpack(i16, {"scale_factor": f64(0.5), "add_offset": f64(1.0)}).dtype == f64
    # an offset without a partner of its type: float64; a scale factor alone: its type
    assert cf_unpack(i16, {"add_offset": f32(1.0)}).dtype == f64
    assert cf_unpack(i16, {"scale_factor": f32(0.5), "add_offset": f64(1.0)}).dtype == f64
    assert cf_unpack(i16, {"scale_factor": f32(0.5)}).dtype == f32
    # no packing attributes, only a fill value: float32 for integers of at most 2 bytes, float64 beyond
    assert cf_unpack(i8, {"_FillValue": np.int8(3)}).dtype == f32
    assert cf_unpack(i16, {"_FillValue": np.int16(3)}).dtype == f32
    assert cf_unpack(i32, {"_FillValue": np.int32(3)}).dtype == f64
    out = cf_unpack(i16, {"scale_factor": f32(0.5), "add_offset": f32(1.0), "_FillValue": np.int16(2)})
    assert np.isnan(out[2]) and out[3] == f32(2.5) and out[0] == f32(1.0)
