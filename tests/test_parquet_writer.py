"""parcels_amd/parquet_writer.py: the multi-threaded writer behind ParticleFile -- files that pyarrow / pandas read back exactly like
the ones pyarrow's own writer produces for the same tables (values, dtypes, field and file metadata), whose bytes do not depend on
the number of encoding threads."""

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest

import parcels_amd as pam
from parcels_amd.parquet_writer import FastParquetWriter, supports_schema


def _cols(n, seed=0):
    rng = np.random.default_rng(seed)
    return {"particle_id": np.arange(n), "t": np.full(n, 3600.0), "x": rng.uniform(0, 360, n).astype(np.float32), "y": rng.uniform(-80, 80, n),
            "state": rng.integers(0, 70, n).astype(np.int32), "flag": rng.random(n) < 0.5, "small": rng.integers(-100, 100, n).astype(np.int16),
            "u8": rng.integers(0, 255, n).astype(np.uint8)}


def _schema(cols):
    return pa.schema([pa.field(k, pa.from_numpy_dtype(v.dtype), metadata={"units": "u_" + k}) for k, v in cols.items()],
                     metadata={"feature_type": "trajectory", "parcels_kernels": "AdvectionRK4"})


@pytest.mark.parametrize("compression", ["zstd", None, "snappy", "lz4", "gzip"])
def test_files_read_back_like_pyarrow_s(tmp_path, compression):
    n = 300_000 if compression != "gzip" else 30_000
    cols, more = _cols(n), _cols(1234, seed=1)
    schema = _schema(cols)
    assert supports_schema(schema)
    fast, ref = str(tmp_path / "fast.parquet"), str(tmp_path / "ref.parquet")
    with FastParquetWriter(fast, schema, compression=compression, row_group_rows=100_000, threads=4) as w:
        w.write_columns(cols)
        w.write_columns({k: v[:0] for k, v in cols.items()})  # an output time nobody reached: nothing is appended
        w.write_columns(more)
    with pq.ParquetWriter(ref, schema, compression=compression or "none", use_dictionary=False) as w:
        w.write_table(pa.table(cols, schema=schema))
        w.write_table(pa.table(more, schema=schema))
    a, b = pq.read_table(fast), pq.read_table(ref)
    assert a.schema.equals(b.schema, check_metadata=True) and a.equals(b)
    md = pq.ParquetFile(fast).metadata
    assert md.num_rows == n + 1234 and md.num_row_groups == -(-n // 100_000) + 1
    assert md.row_group(0).column(0).compression in {None: ("UNCOMPRESSED",), "zstd": ("ZSTD",), "snappy": ("SNAPPY",), "lz4": ("LZ4_RAW", "LZ4"), "gzip": ("GZIP",)}[compression]
    import pandas as pd

    pd.testing.assert_frame_equal(pd.read_parquet(fast), pd.read_parquet(ref))


def test_bytes_do_not_depend_on_the_thread_count(tmp_path):
    cols = _cols(250_000)
    out = []
    for threads in (1, 7):
        p = str(tmp_path / f"t{threads}.parquet")
        with FastParquetWriter(p, _schema(cols), row_group_rows=40_000, threads=threads) as w:
            w.write_columns(cols)
        out.append(open(p, "rb").read())
    assert out[0] == out[1]


def test_particlefile_picks_the_writer(tmp_path):
    import sys, os

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_particlefile_reference import make_fieldset

    fs = make_fieldset()
    pset = pam.ParticleSet(fs, x=np.linspace(0, 1, 50), y=np.zeros(50), t=np.zeros(50))
    pset._data["dt"][:] = 1.0
    files = {}
    for kind in ("auto", "pyarrow", "fast"):
        pf = pam.ParticleFile(tmp_path / f"{kind}.parquet", outputdt=1.0, writer=kind)
        pf.set_metadata("flat")
        with pf:
            pf.write(pset, 0.0)
        files[kind] = pq.read_table(tmp_path / f"{kind}.parquet")
        assert (type(pf).__name__, kind) and files[kind].num_rows == 50
    assert files["auto"].equals(files["pyarrow"]) and files["auto"].schema.equals(files["pyarrow"].schema, check_metadata=True)
    assert open(tmp_path / "auto.parquet", "rb").read() == open(tmp_path / "fast.parquet", "rb").read()
    with pytest.raises(ValueError):
        pam.ParticleFile(tmp_path / "x.parquet", outputdt=1.0, writer="nope")
    with pytest.raises(ValueError, match="writer='fast' needs"):
        pf = pam.ParticleFile(tmp_path / "d.parquet", outputdt=1.0, writer="fast", use_dictionary=True)
        pf.set_metadata("flat")
        pf.write(pset, 0.0)
