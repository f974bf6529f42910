"""The reference's own ParticleFile tests (tests/test_particlefile.py), restated against parcels_amd one for one.

Each test names the reference test it mirrors.  The toy kernels of the reference's tests/common_kernels.py (DoNothing, MoveEast,
MoveNorth) are native kernel tokens here (PK_KERNEL_DO_NOTHING / _MOVE_EAST / _MOVE_NORTH), so the loop runs through the HIP path.
Tests that only touch the host side (ParticleFile construction, ParticleFile.write of host columns, the schema) run without a GPU.
"""

from contextlib import nullcontext as does_not_raise
from datetime import datetime, timedelta

import numpy as np
import pytest

import parcels_amd as pa
from parcels_amd.particlefile import get_schema


def make_fieldset(time="datetime", nt=3, mesh="flat", uniform=None):
    """A small A-grid fieldset like the reference's `fieldset` fixture (tests/conftest.py: ds_2d_left, a datetime time axis)."""
    nx, ny = 12, 10
    md = pa.SGrid2DMetadata(node_dimensions=("XG", "YG"), node_coordinates=("lon", "lat"),
                            face_dimensions=(pa.FaceNodePadding("XC", "XG", pa.Padding.LOW), pa.FaceNodePadding("YC", "YG", pa.Padding.LOW)),
                            vertical_dimensions=None)
    coords = {"lon": (("XG",), np.linspace(-2.0, 4.0, nx)), "lat": (("YG",), np.linspace(-2.0, 3.0, ny))}
    rng = np.random.default_rng(3)
    if time is None:
        dims, shape = ("YG", "XG"), (ny, nx)
    else:
        dims, shape = ("time", "YG", "XG"), (nt, ny, nx)
        if time == "datetime":
            coords["time"] = (("time",), np.datetime64("2000-01-01T00:00:00", "ns") + np.arange(nt) * np.timedelta64(2, "D"))
        else:
            coords["time"] = (("time",), np.arange(nt) * 2 * 86400.0)
    data = {"U": (dims, 1e-5 * rng.standard_normal(shape)), "V": (dims, 1e-5 * rng.standard_normal(shape))}
    if uniform is not None:
        data = {"U": (dims, np.full(shape, uniform[0])), "V": (dims, np.full(shape, uniform[1]))}
    return pa.FieldSet.from_sgrid_conventions(pa.Dataset(data, coords, sgrid=md), mesh=mesh)


@pytest.fixture
def fieldset():
    return make_fieldset()


@pytest.fixture
def tmp_parquet(tmp_path):
    return tmp_path / "tmp.parquet"


# ---- host-side tests (no GPU) ------------------------------------------------------------------------------------------------
def test_pfile_array_remove_particles(fieldset, tmp_parquet):  # test_particlefile.py:76-93
    npart = 10
    pset = pa.ParticleSet(fieldset, x=np.linspace(0, 1, npart), y=0.5 * np.ones(npart), t=fieldset.time_interval.left)
    pfile = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"))
    pset._data["t"][:] = 0
    pfile.write(pset, t=fieldset.time_interval.left)
    pset.remove_indices(3)
    pset._data["t"][:] = 86400
    pfile.write(pset, 86400)
    pfile.close()
    df = pa.read_particlefile(tmp_parquet)
    assert len(df) == 2 * npart - 1
    assert 3 not in df[df["t"] == 86400]["particle_id"].to_numpy()


def test_pfile_array_remove_all_particles(fieldset, tmp_parquet):  # test_particlefile.py:96-114
    npart = 10
    pset = pa.ParticleSet(fieldset, x=np.linspace(0, 1, npart), y=0.5 * np.ones(npart), t=fieldset.time_interval.left)
    pfile = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"))
    pfile.write(pset, t=0)
    for _ in range(npart):
        pset.remove_indices(-1)
    pfile.write(pset, fieldset.time_interval.left + np.timedelta64(1, "D"))
    pfile.write(pset, fieldset.time_interval.left + np.timedelta64(2, "D"))
    pfile.close()
    assert pa.read_particlefile(tmp_parquet)["particle_id"].nunique() == npart


def test_write_dtypes_pfile(fieldset, tmp_parquet):  # test_particlefile.py:117-142
    import pyarrow
    import pyarrow.parquet as pq

    dtypes = [np.float32, np.float64, np.int32, np.uint32, np.int64, np.uint64, np.bool_, np.int8, np.uint8, np.int16, np.uint16]
    MyParticle = pa.get_default_particle(np.float64).add_variable([pa.Variable(f"v_{d.__name__}", dtype=d, initial=0.0) for d in dtypes])
    pset = pa.ParticleSet(fieldset, pclass=MyParticle, x=0, y=0, t=fieldset.time_interval.left)
    pfile = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"))
    pfile.write(pset, t=fieldset.time_interval.left)
    pfile.close()
    tab = pq.read_table(tmp_parquet)
    for d in dtypes:
        assert tab[f"v_{d.__name__}"].type == pyarrow.from_numpy_dtype(d)


@pytest.mark.parametrize("outputdt, expectation", [  # test_particlefile.py:192-206
    (np.timedelta64(5, "s"), does_not_raise()),
    (timedelta(seconds=2), does_not_raise()),
    (5.0, does_not_raise()),
    (np.datetime64("2001-01-02T00:00:00"), pytest.raises(ValueError)),
    (datetime(2000, 1, 2, 0, 0, 0), pytest.raises(ValueError)),
    (-np.timedelta64(5, "s"), pytest.raises(ValueError)),
])
def test_outputdt_types(outputdt, expectation, tmp_parquet):
    from parcels_amd.field import to_seconds

    with expectation:
        pfile = pa.ParticleFile(tmp_parquet, outputdt=outputdt)
        assert pfile.outputdt == to_seconds(outputdt)


def test_particlefile_init(tmp_parquet):  # test_particlefile.py:453-454
    pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"))


def test_particlefile_init_existing_path_no_mode(tmp_parquet):  # test_particlefile.py:476-479
    tmp_parquet.touch()
    with pytest.raises(ValueError, match="already exists"):
        pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"))


def test_particlefile_init_nonexistent_parent(tmp_path):  # test_particlefile.py:482-485
    with pytest.raises(ValueError, match="does not exist"):
        pa.ParticleFile(tmp_path / "nonexistent_dir" / "file.parquet", outputdt=np.timedelta64(1, "s"))


def test_particlefile_init_invalid_mode(tmp_parquet):  # test_particlefile.py:488-490
    with pytest.raises(ValueError, match="Invalid mode value"):
        pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"), mode="something-else")


@pytest.mark.parametrize("name", ["path", "outputdt"])
def test_particlefile_readonly_attrs(tmp_parquet, name):  # test_particlefile.py:493-497
    pfile = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"))
    # "property ... has no setter" is CPython >= 3.11's wording, "can't set attribute" 3.10's (this image)
    with pytest.raises(AttributeError, match="property .* of 'ParticleFile' object has no setter|can't set attribute"):
        setattr(pfile, name, "something")


def test_particlefile_init_invalid(tmp_path):  # test_particlefile.py:500-503
    with pytest.raises(ValueError, match="file extension must be '.parquet'"):
        pa.ParticleFile(tmp_path / "file.not-parquet", outputdt=np.timedelta64(1, "s"))


@pytest.mark.parametrize("spatial", [np.float64, np.float32])
def test_particle_schema(spatial):  # test_particlefile.py:555-601
    import pyarrow

    from parcels_amd.field import TimeInterval

    particle = pa.get_default_particle(spatial)
    s = get_schema(particle, {}, TimeInterval(np.datetime64("2023-01-01T12:00:00"), np.datetime64("2023-01-02T12:00:00")))
    written = [v for v in particle.variables if v.to_write]
    assert len(s.names) == len(written)
    for variable, field in zip(written, s):
        assert variable.name == field.name
        if variable.name != "t":
            assert variable.attrs == {k.decode(): v.decode() for k, v in field.metadata.items()}
        else:
            assert field.metadata[b"units"] == b"seconds since 2023-01-01 12:00:00"
            assert field.metadata[b"calendar"] == b"standard"
        assert pyarrow.from_numpy_dtype(variable.dtype) == field.type


# ---- through the HIP time loop ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_metadata(gpu, fieldset, tmp_parquet):  # test_particlefile.py:33-40
    import pyarrow.parquet as pq

    pset = pa.ParticleSet(fieldset, x=0, y=0)
    ofile = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"))
    pset.execute(pa.DoNothing, runtime=np.timedelta64(1, "s"), dt=np.timedelta64(1, "s"), output_file=ofile)
    tab = pq.read_table(tmp_parquet)
    assert tab.schema.metadata[b"parcels_kernels"].decode().lower() == "DoNothing".lower()
    assert tab.schema.metadata[b"feature_type"] == b"trajectory"


@pytest.mark.gpu
@pytest.mark.parametrize("compression", ["zstd", "gzip", "snappy", "brotli", None])
def test_compression(gpu, fieldset, tmp_parquet, compression):  # test_particlefile.py:43-57
    import pyarrow.parquet as pq

    pset = pa.ParticleSet(fieldset, x=0, y=0)
    ofile = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"), compression=compression)
    pset.execute(pa.DoNothing, runtime=np.timedelta64(1, "s"), dt=np.timedelta64(1, "s"), output_file=ofile)
    tab = pq.ParquetFile(tmp_parquet)
    assert tab.num_row_groups == 2
    for i in range(tab.num_row_groups):
        rg = tab.metadata.row_group(i)
        for j in range(rg.num_columns):
            col = rg.column(j)
            assert col.compression.lower() == compression or (compression is None and col.compression.lower() == "uncompressed")


@pytest.mark.gpu
def test_write_fieldset_without_time(gpu, tmp_parquet):  # test_particlefile.py:60-73
    import pyarrow.parquet as pq

    fieldset = make_fieldset(time=None)
    assert fieldset.time_interval is None
    pset = pa.ParticleSet(fieldset, x=0, y=0)
    ofile = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"))
    pset.execute(pa.DoNothing, runtime=np.timedelta64(1, "s"), dt=np.timedelta64(1, "s"), output_file=ofile)
    table = pq.read_table(tmp_parquet)
    assert table.schema.field("t").metadata[b"units"] == b"seconds"
    assert b"calendar" not in table.schema.field("t").metadata
    assert table["t"].to_numpy()[1] == 1.0


@pytest.mark.gpu
def test_file_warnings(gpu, fieldset, tmp_parquet):  # test_particlefile.py:185-189
    pset = pa.ParticleSet(fieldset, x=[0, 0], y=[0, 0], t=[np.timedelta64(0, "s"), np.timedelta64(1, "s")])
    pfile = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(2, "s"))
    with pytest.warns(pa.ParticleSetWarning, match="Some of the particles have a start time difference.*"):
        pset.execute(pa.AdvectionRK4, runtime=3, dt=1, output_file=pfile)


@pytest.mark.gpu
def test_write_timebackward(gpu, fieldset, tmp_parquet):  # test_particlefile.py:209-219
    release_time = fieldset.time_interval.left + np.array([np.timedelta64(i + 1, "s") for i in range(3)])
    pset = pa.ParticleSet(fieldset, y=[0, 1, 2], x=[0, 0, 0], t=release_time)
    pfile = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"))
    pset.execute(pa.DoNothing, runtime=np.timedelta64(3, "s"), dt=-np.timedelta64(1, "s"), output_file=pfile)
    df = pa.read_particlefile(tmp_parquet)
    assert df["particle_id"].dtype == "int64"
    dt_per_particle = df.groupby("particle_id")["t"].diff().dropna()
    assert len(dt_per_particle) > 0 and (dt_per_particle < 0).all()


@pytest.mark.gpu
def test_reset_dt(gpu, fieldset, tmp_parquet):  # test_particlefile.py:331-344 (Update_lon == MoveEast)
    dt = np.timedelta64(20, "s")
    pset = pa.ParticleSet(fieldset, pclass=pa.get_default_particle(np.float64), x=[0], y=[0])
    ofile = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(50, "s"))
    pset.execute(pa.MoveEast, runtime=5 * dt, dt=dt, output_file=ofile)
    assert np.allclose(pset.x, 0.6)  # steps 20, 20, 10, 20, 20, 10: six kernel executions
    df = pa.read_particlefile(tmp_parquet)
    np.testing.assert_allclose(df["t"].to_numpy(), [0.0, 50.0, 100.0])
    np.testing.assert_allclose(df["x"].to_numpy(), [0.0, 0.3, 0.6])


@pytest.mark.gpu
def test_subsecond_outputdt(gpu, fieldset, tmp_parquet):  # test_particlefile.py:347-362, dt = 100 ms (MoveEast's increment is 0.1)
    pset = pa.ParticleSet(fieldset, x=[0], y=[0])
    ofile = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(100, "ms"))
    pset.execute(pa.MoveEast, runtime=np.timedelta64(1, "s"), dt=np.timedelta64(100, "ms"), output_file=ofile)
    df = pa.read_particlefile(tmp_parquet)
    np.testing.assert_allclose(df["x"], np.arange(0, 1 + 1e-6, 0.1), atol=1e-6)
    np.testing.assert_allclose(df["t"] - df["t"].min(), np.arange(0, 1001, 100) / 1000.0, atol=1e-3)


def _setup_pset_execute(fieldset, outputdt, tmp_path, **execute_kwargs):  # test_particlefile.py:380-398
    npart = 10
    lon, lat = fieldset.U.grid.lon, fieldset.U.grid.lat
    pset = pa.ParticleSet(fieldset, x=np.full(npart, lon.mean()), y=np.full(npart, lat.mean()))
    name = tmp_path / "tmp_exec.parquet"
    pset.execute(pa.DoNothing, output_file=pa.ParticleFile(name, outputdt=outputdt), **execute_kwargs)
    return pa.read_particlefile(name)


@pytest.mark.gpu
def test_pset_execute_outputdt_forwards(gpu, fieldset, tmp_path):  # test_particlefile.py:401-409
    outputdt = timedelta(hours=1)
    df = _setup_pset_execute(fieldset, outputdt, tmp_path, runtime=timedelta(hours=5), dt=timedelta(minutes=5))
    np.testing.assert_equal(np.diff(df[df["particle_id"] == 0]["t"]), outputdt.seconds)


@pytest.mark.gpu
def test_pset_execute_output_time_forwards(gpu, fieldset, tmp_path):  # test_particlefile.py:412-420
    runtime = np.timedelta64(5, "h")
    df = _setup_pset_execute(fieldset, np.timedelta64(1, "h"), tmp_path, runtime=runtime, dt=np.timedelta64(5, "m"))
    assert df["t"].min() == 0.0  # seconds since fieldset.time_interval.left
    assert df["t"].max() - df["t"].min() == runtime / np.timedelta64(1, "s")


@pytest.mark.gpu
@pytest.mark.parametrize("time", ["datetime", None])
def test_pset_execute_outputdt_backwards(gpu, tmp_path, time):  # test_particlefile.py:423-450 (static and time-varying fieldsets)
    outputdt = timedelta(hours=1)
    df = _setup_pset_execute(make_fieldset(time=time), outputdt, tmp_path, runtime=timedelta(days=2), dt=-timedelta(minutes=5))
    t0 = df[df["particle_id"] == 0]["t"].to_numpy()
    assert len(t0) == 49
    np.testing.assert_equal(np.diff(t0), -outputdt.seconds)


@pytest.mark.gpu
def test_particlefile_init_existing_path_modes(gpu, fieldset, tmp_parquet):  # test_particlefile.py:457-473
    pset = pa.ParticleSet(fieldset, x=0, y=0)
    first = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"))
    pset.execute(pa.DoNothing, runtime=np.timedelta64(10, "s"), dt=np.timedelta64(1, "s"), output_file=first)
    df_first = pa.read_particlefile(tmp_parquet)
    with pytest.raises(ValueError, match="already exists"):
        pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"))
    overwrite = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"), mode="w")
    pset.execute(pa.DoNothing, runtime=np.timedelta64(10, "s"), dt=np.timedelta64(1, "s"), output_file=overwrite)
    assert len(df_first) == len(pa.read_particlefile(tmp_parquet)) == 11


@pytest.mark.gpu
@pytest.mark.parametrize("async_output", [True, False])
def test_particlefile_readable_after_kernel_error(gpu, fieldset, tmp_parquet, async_output):  # test_particlefile.py:512-526 (GH-2713)
    """The reference's ErrorKernel sets StatusCode.Error; here the error is the one the hot path raises itself: a particle advected
    out of the domain.  The file must hold a footer and every table written before the error."""
    fs = make_fieldset(uniform=(1.0, 0.0))  # 1 degree/s eastwards on a flat mesh: out of the 6-degree domain within a few steps
    pset = pa.ParticleSet(fs, x=np.zeros(4), y=np.zeros(4))
    pset.async_output = async_output
    ofile = pa.ParticleFile(tmp_parquet, outputdt=np.timedelta64(1, "s"))
    with pytest.raises(pa.FieldOutOfBoundError):
        pset.execute(pa.AdvectionRK4, runtime=np.timedelta64(10, "s"), dt=np.timedelta64(1, "s"), output_file=ofile)
    df = pa.read_particlefile(tmp_parquet)
    assert len(df) >= 4  # at least the initial condition was written
    assert df["t"].max() >= 2.0  # and the tables of the intervals that completed
