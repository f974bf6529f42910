"""parcels_amd/hdf5.py (the HDF5 / classic-NetCDF reader behind NetCDFLevels) against files written by the HDF5 library itself --
tests/golden/hdf5/*, generated in the build container by tools/make_hdf5_fixtures.sh (libhdf5 1.10 of /opt/conda; h5repack of the
reference's own NetCDF-4 sample) -- and against classic NetCDF files written here by scipy.io.netcdf_file.  Every value of the
fixtures is a formula of its indices (tools/make_hdf5_fixtures.c)."""

from __future__ import annotations

import os

import numpy as np
import pytest

import parcels_amd as pa
from parcels_amd.hdf5 import HDF5File, NetCDF3File

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hdf5")
NT, NZ, NY, NX = 5, 3, 6, 8
T, Z, Y, X = np.meshgrid(np.arange(NT), np.arange(NZ), np.arange(NY), np.arange(NX), indexing="ij")
F32 = (1000.0 * T + 100.0 * Z + 10.0 * Y + X + 0.25).astype(np.float32)
PACKED = ((7 * T + 5 * Z + 3 * Y + X) % 2000 - 1000).astype(np.int16)


@pytest.mark.parametrize("name, superblock, index", [("default_v0.h5", 0, "btree1"), ("netcdf4_style_dense.nc", 0, "btree1"), ("libver_latest.h5", 3, "fixed_array")])
def test_reads_what_libhdf5_wrote(name, superblock, index):
    """superblock 0 with symbol-table groups and v1 object headers / creation-order tracking with dense link storage (24 links in
    a fractal heap, what netCDF-4 files with more than 8 variables have) / libver latest (v2 headers, layout v4 chunk indices)."""
    f = HDF5File(os.path.join(HERE, name))
    assert f.sb_version == superblock
    ds = f.datasets()
    assert len(ds) == 23 and "grp/inner" in ds and "extra_variable_number_13" in ds
    u, v, w = ds["U"], ds["V"], ds["W"]
    assert u.layout["class"] == "chunked" and u.layout["index"] == index and u.layout["chunk"] == (1, NZ, 4, 5)
    assert [fid for fid, _ in u.filters] == [2, 1] and [fid for fid, _ in v.filters] == [1, 3]  # shuffle + deflate; deflate + fletcher32
    assert np.array_equal(u.read(), F32) and u.attrs["some_number"] == 42.5
    expect_v = PACKED.copy()
    expect_v[3] = -32767  # that time level was never written: the fill value of the dataset
    assert v.dtype == np.int16 and np.array_equal(v.read(), expect_v)
    assert v.attrs["scale_factor"] == 0.01 and v.attrs["add_offset"] == 1.5 and v.attrs["_FillValue"] == -32767
    assert w.dtype == np.dtype(">f8") and w.layout["class"] == "contiguous" and np.array_equal(w.read(), -F32.astype(np.float64))
    assert ds["T2"].layout.get("index") == ("single" if superblock == 3 else "btree1") and np.array_equal(ds["T2"].read(), F32[:, 0])
    assert ds["S"].layout.get("index") == ("implicit" if superblock == 3 else "btree1") and np.array_equal(ds["S"].read(), F32)
    assert np.array_equal(ds["F"].read(), F32)
    for k in range(NT):  # one time level = the chunks that intersect it, nothing else
        for d, full in ((u, F32), (v, expect_v), (w, -F32.astype(np.float64)), (ds["S"], F32), (ds["F"], F32)):
            assert np.array_equal(d.read(first=k), full[k])
    assert np.array_equal(ds["time_counter"].read(), 86400.0 * np.arange(NT))
    assert np.array_equal(ds["extra_variable_number_07"].read(), 7 + 0.125 * np.arange(NX))
    with pytest.raises(IndexError):
        u.read(first=NT)


def test_reference_sample_repacked_by_h5repack():
    """The reference's NetCDF-4 regression input (tests/test_data/test_interpolation_data_random_linear.nc: superblock 2, contiguous)
    re-packed to one chunk per time level with deflate: same values as the golden fixture decoded from the original."""
    f = HDF5File(os.path.join(HERE, "reference_linear_chunked_gzip.nc"))
    g = np.load(os.path.join(os.path.dirname(HERE), "v3jit_linear.npz"))
    for nm in "UVW":
        d = f.dataset(nm)
        assert d.layout["class"] == "chunked" and d.layout["chunk"] == (1, 5, 10, 10) and [fid for fid, _ in d.filters] == [1]
        if nm in g:
            assert np.array_equal(d.read(), g[nm])
    src = pa.NetCDFLevels(os.path.join(HERE, "reference_linear_chunked_gzip.nc"), "U")
    assert src.shape == (20, 5, 10, 10) and src.dtype == np.float64
    assert np.array_equal(src.level(7), f.dataset("U").read()[7])
    assert np.array_equal(pa.read_netcdf_variable(os.path.join(HERE, "reference_linear_chunked_gzip.nc"), "depth"), f.dataset("depth").read())


def test_netcdf_levels_unpack_cf_and_concatenate_files():
    p = os.path.join(HERE, "netcdf4_style_dense.nc")
    v = pa.NetCDFLevels(p, "V")
    assert v.shape == (NT, NZ, NY, NX) and v.dtype == np.float64  # float64 packing attributes -> float64 values
    assert np.array_equal(v.read_level(2), PACKED[2].astype(np.float64) * 0.01 + 1.5)
    assert np.isnan(v.read_level(3)).all() and np.all(v.level(3) == 0.0)  # _FillValue -> NaN -> 0 (model.py:135-143)
    two = pa.NetCDFLevels([os.path.join(HERE, "default_v0.h5"), os.path.join(HERE, "libver_latest.h5")], "U")  # files continuing each other in time
    assert two.shape == (2 * NT, NZ, NY, NX) and two.dtype == np.float32
    assert np.array_equal(two.level(1), F32[1]) and np.array_equal(two.level(NT + 3), F32[3])
    surf = pa.NetCDFLevels(p, "T2")  # (time, y, x) -> (nt, 1, ny, nx)
    assert surf.shape == (NT, 1, NY, NX) and np.array_equal(surf.level(4)[0], F32[4, 0])
    with pytest.raises(KeyError):
        pa.NetCDFLevels(p, "no_such_variable")


@pytest.mark.parametrize("version", [1, 2])
def test_classic_netcdf3_files(tmp_path, version):
    """CDF-1 / CDF-2 written by scipy.io.netcdf_file: a record (unlimited time) variable interleaved with a second one, a fixed
    variable, packed shorts with CF attributes."""
    from scipy.io import netcdf_file

    p = str(tmp_path / "classic.nc")
    with netcdf_file(p, "w", version=version) as nc:
        nc.createDimension("time", None)
        nc.createDimension("z", NZ)
        nc.createDimension("y", NY)
        nc.createDimension("x", NX)
        u = nc.createVariable("U", "f4", ("time", "z", "y", "x"))
        s = nc.createVariable("S", "i2", ("time", "y", "x"))
        s.scale_factor = np.float32(0.5)
        s.add_offset = np.float32(10.0)
        s._FillValue = np.int16(-999)
        lon = nc.createVariable("lon", "f8", ("x",))
        lon[:] = 0.25 * np.arange(NX)
        for k in range(NT):
            u[k] = F32[k]
            s[k] = PACKED[k, 0]
        s[1, 2, 3] = -999
    f = NetCDF3File(p)
    assert f.numrecs == NT and f.shape("U") == (NT, NZ, NY, NX)
    assert np.array_equal(f.read("U"), F32) and np.array_equal(f.read("U", 3), F32[3]) and np.array_equal(f.read("lon"), 0.25 * np.arange(NX))
    src = pa.NetCDFLevels(p, "S")
    assert src.shape == (NT, 1, NY, NX) and src.dtype == np.float32
    want = PACKED[1, 0].astype(np.float32) * np.float32(0.5) + np.float32(10.0)
    got = src.read_level(1)[0]
    assert np.isnan(got[2, 3]) and np.array_equal(np.delete(got.ravel(), 2 * NX + 3), np.delete(want.ravel(), 2 * NX + 3))
    assert np.array_equal(pa.NetCDFLevels(p, "U").level(4), F32[4])


def test_classic_netcdf3_long_header_and_streaming_record_count(tmp_path):
    """A header beyond the first 4 MiB (many / long attributes) is parsed from a longer prefix, and the "streaming" record count
    0xFFFFFFFF (a writer that never went back to fill it in) is replaced by what the file size holds."""
    from scipy.io import netcdf_file

    p = str(tmp_path / "long.nc")
    with netcdf_file(p, "w", version=2) as nc:
        nc.createDimension("time", None)
        nc.createDimension("x", NX)
        nc.history = "h" * (5 << 20)  # 5 MiB of global attribute in FRONT of the variable list
        u = nc.createVariable("U", "f4", ("time", "x"))
        for k in range(NT):
            u[k] = F32[k, 0, 0]
    f = NetCDF3File(p)
    assert f.numrecs == NT and np.array_equal(f.read("U"), F32[:, 0, 0])
    f.close()
    raw = bytearray(open(p, "rb").read())
    raw[4:8] = b"\xff\xff\xff\xff"
    open(p, "wb").write(bytes(raw))
    g = NetCDF3File(p)
    assert g.numrecs == NT and np.array_equal(g.read("U", NT - 1), F32[NT - 1, 0, 0])
    g.close()
    # a header whose LAST item is a byte slice that straddles the prefix (no variables behind the attribute): slices do not raise
    # when cut short, the parse offset beyond the prefix is what asks for the longer read
    p2 = str(tmp_path / "attr_only.nc")
    with netcdf_file(p2, "w", version=2) as nc:
        nc.createDimension("x", NX)
        nc.history = "h" * (5 << 20)
    h = NetCDF3File(p2)
    assert len(h.gattrs["history"]) == 5 << 20
    h.close()
    q = tmp_path / "cut.nc"
    q.write_bytes(bytes(raw[:1000]))
    with pytest.raises(ValueError, match="truncated or malformed"):
        NetCDF3File(str(q))


def test_unsupported_features_say_so(tmp_path):
    with pytest.raises(ValueError, match="not an HDF5 file"):
        p = tmp_path / "x.h5"
        p.write_bytes(b"not hdf5 at all" * 100)
        HDF5File(str(p))
