"""CPU: the trajectory files of parcels_amd.ParticleFile against those of the reference's REAL ParticleFile (src/parcels/_core/particlefile.py
loaded under oracle/ref_shim.py; its writer needs only pyarrow, which is here): the same particle columns written at the same output times
give the same Parquet schema (names, Arrow types, per-column CF attributes, file metadata), the same row groups and the same values --
through pyarrow's writer and through the multi-threaded writer of parcels_amd/parquet_writer.py alike."""
import importlib

import numpy as np
import pytest

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")


def _reference_particlefile():
    m = ref_shim.load_reference()
    import sys

    sys.modules["parcels"].__version__ = "reference"
    return m, importlib.import_module("parcels._core.particlefile")


def _sets(n=40, seed=3):
    import parcels_amd as pa
    from case_utils import build_fieldset
    from oracle import cases
    from oracle.make_golden import build_ref_fieldset

    m, _ = _reference_particlefile()
    case = cases.rect_agrid_case("pf", mesh="spherical", kernels=["AdvectionRK4"], seed=1, npart=4, nx=6, ny=5, nz=2, nt=3)
    ref_fs, my_fs = build_ref_fieldset(case)[0], build_fieldset(case)
    RP = m["particle"]
    rclass = RP.get_default_particle(np.float32).add_variable([
        RP.Variable("age", dtype=np.float32, initial=0, attrs={"units": "s", "long_name": "age of the particle"}),
        RP.Variable("hidden", dtype=np.float64, initial=1, to_write=False), RP.Variable("count", dtype=np.int32, initial=0)])
    mclass = pa.get_default_particle(np.float32).add_variable([
        pa.Variable("age", dtype=np.float32, initial=0, attrs={"units": "s", "long_name": "age of the particle"}),
        pa.Variable("hidden", dtype=np.float64, initial=1, to_write=False), pa.Variable("count", dtype=np.int32, initial=0)])
    rng = np.random.default_rng(seed)
    x, y = rng.uniform(0.5, 3.0, n), rng.uniform(0.5, 2.0, n)
    t0 = np.where(np.arange(n) % 5 == 0, 600.0, 0.0)
    rset = m["particleset"].ParticleSet(ref_fs, pclass=rclass, x=x, y=y, z=np.zeros(n), t=(t0 * 1e9).astype("int64").astype("timedelta64[ns]"))
    mset = pa.ParticleSet(my_fs, pclass=mclass, x=x, y=y, z=np.zeros(n), t=t0)
    return rset, mset, ref_fs, my_fs


def _advance(sets, rng_seed, step):
    """The same change of the columns in both sets (what an output interval of a run does to them)."""
    for s in sets:
        rng = np.random.default_rng(rng_seed)
        d = s._data
        n = len(d["t"])
        d["t"][:] = d["t"] + np.where(np.isfinite(d["t"]), step, 0.0)
        d["x"][:] = d["x"] + rng.normal(scale=0.01, size=n).astype(d["x"].dtype)
        d["age"][:] = d["age"] + np.float32(step)
        d["count"][:] = d["count"] + 1
        d["dt"][:] = step


@pytest.mark.parametrize("writer", ["pyarrow", "fast"])
@pytest.mark.parametrize("compression", ["zstd", None])
def test_same_file_as_the_reference(tmp_path, writer, compression):
    import pyarrow.parquet as pq

    import parcels_amd as pa

    _, rpf_mod = _reference_particlefile()
    rset, mset, ref_fs, my_fs = _sets()
    rpath, mpath = tmp_path / "ref.parquet", tmp_path / "mine.parquet"
    rpf = rpf_mod.ParticleFile(rpath, outputdt=600.0, compression=compression)
    mpf = pa.ParticleFile(mpath, outputdt=600.0, compression=compression, writer=writer, distributed=False)
    rpf.set_metadata(ref_fs.gridset[0]._mesh)
    mpf.set_metadata(my_fs.gridset[0]._mesh)
    for s in (rset, mset):
        s._data["dt"][:] = 600.0
    t = 0.0
    for k in range(4):  # output times 0, 600, 1200, 1800: the particles released at 600 s join at the second one
        rpf.write(rset, t)
        mpf.write(mset, t)
        _advance([s for s in (rset, mset)], 10 + k, 600.0)
        if k == 1:  # some particles are deleted between two output times
            for s in (rset, mset):
                s.remove_indices(np.array([3, 7, 11]))
        if k == 0:  # the late particles have not started yet: their clock stands still
            for s in (rset, mset):
                late = s._data["particle_id"] % 5 == 0
                s._data["t"][late] = 600.0
        t += 600.0
    rpf.close()
    mpf.close()
    a, b = pq.read_table(rpath), pq.read_table(mpath)
    assert a.schema.names == b.schema.names
    for fa, fb in zip(a.schema, b.schema):
        assert fa.type == fb.type, fa.name
        assert (fa.metadata or {}) == (fb.metadata or {}), fa.name
    ma, mb = dict(a.schema.metadata), dict(b.schema.metadata)
    for m_ in (ma, mb):
        m_.pop(b"parcels_version", None)
        m_.pop(b"ARROW:schema", None)
    assert ma == mb
    assert a.num_rows == b.num_rows > 0
    for name in a.schema.names:
        assert np.array_equal(a.column(name).to_numpy(), b.column(name).to_numpy(), equal_nan=True), name
    fa, fb = pq.ParquetFile(rpath), pq.ParquetFile(mpath)
    assert fa.metadata.num_row_groups == fb.metadata.num_row_groups == 4
    assert [fa.metadata.row_group(i).num_rows for i in range(4)] == [fb.metadata.row_group(i).num_rows for i in range(4)]
    assert "hidden" not in a.schema.names and {"age", "count", "x", "t", "particle_id"} <= set(a.schema.names)


def test_constructor_refuses_what_the_reference_refuses(tmp_path):
    import parcels_amd as pa

    _, rpf_mod = _reference_particlefile()
    (tmp_path / "exists.parquet").write_bytes(b"x")
    for args, kw in ((("out.parquet", 3), {}), (("out.zarr", 600.0), {}), (("out.parquet", -1.0), {}), (("out.parquet", 600.0), {"mode": "a"}),
                     (("exists.parquet", 600.0), {}), (("nowhere/out.parquet", 600.0), {})):
        out = []
        for cls, extra in ((rpf_mod.ParticleFile, {}), (pa.ParticleFile, {"distributed": False})):
            try:
                cls(tmp_path / args[0], args[1], **kw, **extra)
                out.append(("ok", ""))
            except Exception as e:  # noqa: BLE001
                out.append((type(e).__name__, str(e)))
        assert out[0] == out[1] and out[0][0] != "ok", (args, kw, out)
    ok = pa.ParticleFile(tmp_path / "exists.parquet", 600.0, mode="w", distributed=False)
    assert not (tmp_path / "exists.parquet").exists() and ok.outputdt == rpf_mod.ParticleFile(tmp_path / "new.parquet", np.timedelta64(10, "m")).outputdt
