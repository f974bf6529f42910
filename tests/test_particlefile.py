"""ParticleFile (Parquet) -- host-side write-out (mirrors the reference's tests/test_particlefile.py patterns)."""

import numpy as np
import pytest

import parcels_amd as pa
from case_utils import build_fieldset, build_pset, load_golden


def _pset():
    case, _, _ = load_golden("agrid_sph_rk4_f64")
    fs = build_fieldset(case)
    return case, fs, build_pset(case, fs)


def test_validation(tmp_path):
    with pytest.raises(ValueError):
        pa.ParticleFile(tmp_path / "a.zarr", outputdt=3600.0)
    with pytest.raises(ValueError):
        pa.ParticleFile(tmp_path / "a.parquet", outputdt=0.0)
    with pytest.raises(ValueError):
        pa.ParticleFile(tmp_path / "a.parquet", outputdt=3600)  # int is rejected like the reference
    (tmp_path / "b.parquet").write_bytes(b"x")
    with pytest.raises(ValueError):
        pa.ParticleFile(tmp_path / "b.parquet", outputdt=3600.0)
    pa.ParticleFile(tmp_path / "b.parquet", outputdt=3600.0, mode="w")


def test_write_filter_and_schema(tmp_path):
    case, fs, pset = _pset()
    n = len(pset)
    pset._data["dt"][:] = 3600.0
    pset._data["t"][: n // 2] = 7200.0  # half of the particles are at the output time
    pset._data["t"][n // 2 :] = 0.0
    with pa.ParticleFile(tmp_path / "out.parquet", outputdt=3600.0) as pf:
        pf.write(pset, 7200.0)
        pf.write(pset, 0.0)
        pf.write(pset, 3600.0 * 5)  # nobody
    df = pa.read_particlefile(tmp_path / "out.parquet")
    assert list(df.columns) == ["t", "z", "y", "x", "particle_id"]  # to_write variables only (particle.py:129-175)
    assert len(df) == n
    first = df.iloc[: n // 2]
    assert np.all(first["t"] == 7200.0) and np.array_equal(first["particle_id"], np.arange(n // 2))
    assert df["x"].dtype == np.float64 and df["particle_id"].dtype == np.int64


@pytest.mark.gpu
def test_execute_with_output_file(gpu, tmp_path):
    case, fs, pset = _pset()
    n = len(pset)
    pf = pa.ParticleFile(tmp_path / "traj.parquet", outputdt=6 * 3600.0)
    pset.execute(pa.AdvectionRK4, dt=3600.0, runtime=24 * 3600.0, output_file=pf)
    df = pa.read_particlefile(tmp_path / "traj.parquet")
    assert len(df) == 5 * n  # t = 0, 6, 12, 18, 24 h
    assert sorted(df["t"].unique()) == [0.0, 21600.0, 43200.0, 64800.0, 86400.0]
    last = df[df["t"] == 86400.0].sort_values("particle_id")
    np.testing.assert_array_equal(last["x"].to_numpy(), pset.x)
    # same trajectory as one uninterrupted execute
    _, _, p2 = _pset()
    p2.execute(pa.AdvectionRK4, dt=3600.0, runtime=24 * 3600.0)
    np.testing.assert_array_equal(p2.x, pset.x)


@pytest.mark.gpu
def test_output_with_deletions_matches_uninterrupted_run(gpu, tmp_path):
    """Device-resident loop + write-out + deletions: the file holds the survivors of each output time and the final
    positions equal a run without output (tests/test_particlefile.py delete+write pattern of the reference)."""
    case, _, _ = load_golden("agrid_flat_rk4_3d_escape_delete")
    kernels = [pa.AdvectionRK4_3D, pa.DeleteParticle]
    fs = build_fieldset(case)
    a = build_pset(case, fs)
    a.execute(kernels, dt=case["dt"], runtime=case["runtime"])
    fs2 = build_fieldset(case)
    b = build_pset(case, fs2)
    pf = pa.ParticleFile(tmp_path / "del.parquet", outputdt=float(2 * case["dt"]))
    b.execute(kernels, dt=case["dt"], runtime=case["runtime"], output_file=pf)
    assert len(a) == len(b) < len(np.atleast_1d(case["x"]))
    np.testing.assert_array_equal(a.particle_id, b.particle_id)
    np.testing.assert_array_equal(a.x, b.x)
    np.testing.assert_array_equal(a.z, b.z)
    df = pa.read_particlefile(tmp_path / "del.parquet")
    counts = df.groupby("t").size()
    assert counts.iloc[0] == len(np.atleast_1d(case["x"])) and counts.iloc[-1] == len(b)
    assert np.all(np.diff(counts.to_numpy()) <= 0)  # particles only ever disappear
    last = df[df["t"] == df["t"].max()].sort_values("particle_id")
    np.testing.assert_array_equal(last["particle_id"].to_numpy(), b.particle_id)
    np.testing.assert_array_equal(last["x"].to_numpy(), b.x)


@pytest.mark.gpu
@pytest.mark.parametrize("sort", [False, True])
def test_async_write_out_equals_the_inline_path(gpu, tmp_path, sort):
    """The double-buffered write-out (device snapshot -> D2H on the copy stream -> filter + Parquet encode on a writer thread,
    behind the next interval's launch) produces the file of the inline path byte for byte: deletions between intervals, the cell
    sort, staggered releases (rows filtered out at some output times) and a host-only user Variable included."""
    from oracle import cases

    case = cases.rect_agrid_case("pf_async", mesh="flat", kernels=["AdvectionRK4", "DeleteParticle"], seed=5, nx=24, ny=16, nz=5, nt=4, npart=20000,
                                 vel=2.5, margin=0.02, runtime=30 * 3600.0, stagger=True)
    files = {}
    for mode in ("inline", "async"):
        fs = build_fieldset(case)
        pclass = pa.get_default_particle(np.float64).add_variable(pa.Variable("tag", dtype=np.int32, initial=0))
        n = len(case["x"])
        pset = pa.ParticleSet(fs, pclass=pclass, x=case["x"], y=case["y"], z=case["z"], t=case["t0"], sort_by_cell=sort, tag=np.arange(n) % 13)
        pset.async_output = mode == "async"
        path = tmp_path / f"{mode}_{sort}.parquet"
        pset.execute([pa.AdvectionRK4, pa.kernels.DeleteParticle], dt=3600.0, runtime=case["runtime"], output_file=pa.ParticleFile(path, outputdt=2 * 3600.0))
        files[mode] = (path.read_bytes(), len(pset), {k: np.array(v) for k, v in pset._data.items()})
    assert files["async"][1] == files["inline"][1] < 20000, "no deletions: the test does not test"
    assert files["async"][0] == files["inline"][0]
    for k, v in files["inline"][2].items():
        assert np.array_equal(files["async"][2][k], v), k
    df = pa.read_particlefile(tmp_path / f"async_{sort}.parquet")
    assert "tag" in df.columns and len(df["t"].unique()) >= 16


def test_take_rows_shortcut_only_for_the_identity_selection():
    """particlefile.py:142-180 indexes particle_data[v][indices]: an index array that merely LOOKS like `all rows` (same length, starts
    at 0, ends at n - 1) but permutes them must permute the columns."""
    from parcels_amd.particlefile import _take_rows

    data = {"a": np.arange(4.0), "b": np.arange(4) * 10}
    same = _take_rows(data, ["a", "b"], np.arange(4))
    assert same["a"] is data["a"] and same["b"] is data["b"]  # the columns themselves, no copy
    perm = _take_rows(data, ["a", "b"], np.array([0, 2, 1, 3]))
    assert perm["a"].tolist() == [0.0, 2.0, 1.0, 3.0] and perm["b"].tolist() == [0, 20, 10, 30]


def test_async_writer_copies_host_only_variables_of_hosted_kernel_lists():
    """A Python kernel on the host path updates user Variables IN PLACE during the next interval (hostkernels.execute_hosted), while
    the writer thread still encodes the previous output time: its table must hold that output time's values."""
    import threading

    from parcels_amd.particlefile import _AsyncWriter

    gate = threading.Event()
    written = []

    class Engine:
        _SNAP_COLS = ("t", "x")
        device_variables = []

        def snapshot_begin(self, cols, slot):
            pass

        def snapshot_wait(self, slot):
            gate.wait(5)  # the encode of this table starts only after the caller has gone on
            return {"t": np.zeros(3), "x": np.zeros(3)}

    class File:
        def write(self, view, t):
            written.append((t, view._data["age"].copy()))

    class Kern:
        host_functions = ["Age"]

    class PSet:
        _pclass, fieldset, _kernel = None, None, Kern()

    w = _AsyncWriter(File(), PSet(), Engine(), ["t", "x", "age"])
    age = np.array([1.0, 2.0, 3.0])
    w.submit({"t": np.zeros(3), "x": np.zeros(3), "age": age}, 10.0)
    age += 100.0  # the next interval's Python kernel, in place
    gate.set()
    w.close()
    assert written and written[0][0] == 10.0 and written[0][1].tolist() == [1.0, 2.0, 3.0]
    # a device kernel list never mutates host-only columns: no copy is made for it
    Kern.host_functions = []
    gate.clear()
    w2 = _AsyncWriter(File(), PSet(), Engine(), ["t", "x", "age"])
    age2 = np.array([5.0, 6.0, 7.0])
    w2.submit({"t": np.zeros(3), "x": np.zeros(3), "age": age2}, 20.0)
    gate.set()
    w2.close()
    assert written[-1][1].tolist() == [5.0, 6.0, 7.0]


def test_async_writer_two_tables_in_flight_commit_in_submission_order():
    """Round 6: two writer threads -- table k+1 prepares (compresses) while table k commits (file writes); the commits happen in submission
    order whatever the threads' timing, and a table that fails passes its turn on instead of blocking the ones behind it."""
    import threading
    import time
    import types

    from parcels_amd.particlefile import _AsyncWriter

    log, lock = [], threading.Lock()

    class Engine:
        _SNAP_COLS = ("t", "dt", "x")
        device_variables = []

        def snapshot_begin(self, cols, slot, filter_t=None):
            assert filter_t is not None  # every to-write Variable is a device column: the filter runs on the device

        def snapshot_wait(self, slot):
            return {"t": np.zeros(2), "dt": np.ones(2), "x": np.zeros(2)}

    class File:
        _collective = False

        def prepare_columns(self, pclass, cols, ti):
            k = len([e for e in log if e[0] == "prepare"])
            with lock:
                log.append(("prepare", k))
            if k == 0:
                time.sleep(0.3)  # the FIRST table is slow to prepare: the second one is ready to commit long before it
            if k == 2:
                raise RuntimeError("table 2 cannot be encoded")
            return k

        def commit_columns(self, ticket):
            with lock:
                log.append(("commit", ticket))

    pclass = types.SimpleNamespace(variables=[types.SimpleNamespace(name="t", to_write=True), types.SimpleNamespace(name="x", to_write=True)])
    pset = types.SimpleNamespace(_pclass=pclass, fieldset=types.SimpleNamespace(time_interval=None), _kernel=None)
    w = _AsyncWriter(File(), pset, Engine(), ["t", "dt", "x"])
    data = {"t": np.zeros(2), "dt": np.ones(2), "x": np.zeros(2)}
    for k in range(4):
        try:
            w.submit(data, float(k))
        except RuntimeError:
            pass  # (the failure of table 2 surfaces when its slot is reused)
    try:
        w.close()
    except RuntimeError:
        pass
    commits = [e[1] for e in log if e[0] == "commit"]
    assert commits == [0, 1, 3], log  # in order, table 2 skipped, table 3 not stuck behind it
    assert log.index(("prepare", 1)) < log.index(("commit", 0)), log  # table 1 prepared while table 0 was still busy
