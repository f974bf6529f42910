"""CPU: the multi-rank path (sharding + write-out all-gather) with world_size 2 over gloo."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from parcels_amd.distributed import gather_output_columns, shard_slice


def test_shard_slice_partitions_exactly():
    for n in (0, 1, 7, 8, 1000, 10_000_001):
        for world in (1, 2, 3, 8):
            sl = [shard_slice(n, r, world) for r in range(world)]
            assert sl[0].start == 0 and sl[-1].stop == n
            assert all(sl[i].stop == sl[i + 1].start for i in range(world - 1))
            sizes = [s.stop - s.start for s in sl]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sl = shard_slice(n_total, rank, world)
        ids = np.arange(n_total, dtype=np.int64)[sl]
        # ragged: rank 1 "deleted" a few particles
        if rank == 1:
            ids = ids[:-3]
        cols = {
            "particle_id": torch.from_numpy(ids.copy()),
            "x": torch.from_numpy(ids.astype(np.float64) * 0.5),
            "t": torch.full((len(ids),), 3600.0, dtype=torch.float64),
            "z": torch.from_numpy(ids.astype(np.float32)),
        }
        out = gather_output_columns(cols)
        if rank == 0:
            q.put({k: v.numpy() for k, v in out.items()})
    finally:
        dist.destroy_process_group()


def test_write_out_allgather_world2():
    n_total, world = 1001, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect_ids = np.arange(n_total - 3, dtype=np.int64)
    assert np.array_equal(out["particle_id"], expect_ids)
    assert np.array_equal(out["x"], expect_ids * 0.5)
    assert out["z"].dtype == np.float32 and np.array_equal(out["z"], expect_ids.astype(np.float32))
    assert np.all(out["t"] == 3600.0)


# ---- the sharded ParticleSet + collective ParticleFile.write: two ranks produce the single-process file, byte for byte --------
def _synthetic_run(pset, pf, times):
    """Stand-in for ParticleSet.execute's output loop (particleset.py:436-459) on the host columns: positions are a function
    of (particle_id, time); some particles lag behind the output time (filtered out by |t_p - t| <= |dt|/2), some are deleted on
    the way, and one write finds nobody."""
    import numpy as np

    d = pset._data
    d["dt"][:] = 600.0
    for k, tm in enumerate(times):
        ids = d["particle_id"].astype(np.float64)
        d["x"][:] = (ids * 0.25 + tm * 1e-3).astype(d["x"].dtype)
        d["y"][:] = (np.sin(ids) * 10 + k).astype(d["y"].dtype)
        d["z"][:] = (ids % 7).astype(d["z"].dtype)
        d["t"][:] = tm
        d["t"][d["particle_id"] % 5 == k % 5] = tm - 1000.0  # outside dt/2: not written this time
        if k == 2:
            d["t"][:] = tm + 5000.0  # nobody passes the filter
        pf.write(pset, tm)
        gone = np.where(d["particle_id"] % 11 == k)[0]  # Kernel.remove_deleted between intervals
        pset.remove_indices(gone)


def _make_fieldset():
    import parcels_amd as pa
    from case_utils import build_fieldset
    from oracle import cases

    return build_fieldset(cases.rect_agrid_case("dist", mesh="spherical", kernels=["AdvectionRK4"], seed=1, npart=4))


def _pf_worker(rank, world, port, n_total, path):
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import parcels_amd as pa

        fs = _make_fieldset()
        ids = np.arange(n_total)
        pset = pa.ParticleSet(fs, x=ids * 0.1, y=ids * 0.0, z=ids * 0.0, t=np.zeros(n_total), shard="auto")
        assert pset._shard == (rank, world) and len(pset) == shard_slice(n_total, rank, world).stop - shard_slice(n_total, rank, world).start
        pf = pa.ParticleFile(path, outputdt=600.0)
        pf.set_metadata("spherical")
        with pf:
            _synthetic_run(pset, pf, [0.0, 600.0, 1200.0, 1800.0, 2400.0])
        assert (rank == 0) == os.path.exists(path) or rank != 0
    finally:
        dist.destroy_process_group()


def test_two_ranks_write_the_single_process_file_byte_for_byte(tmp_path):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import parcels_amd as pa

    n_total, world = 1003, 2
    # single process, whole id space
    fs = _make_fieldset()
    ids = np.arange(n_total)
    pset = pa.ParticleSet(fs, x=ids * 0.1, y=ids * 0.0, z=ids * 0.0, t=np.zeros(n_total))
    one = tmp_path / "one.parquet"
    pf = pa.ParticleFile(one, outputdt=600.0)
    pf.set_metadata("spherical")
    with pf:
        _synthetic_run(pset, pf, [0.0, 600.0, 1200.0, 1800.0, 2400.0])
    # two ranks over gloo
    two = tmp_path / "two.parquet"
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_pf_worker, args=(r, world, port, n_total, str(two))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert one.read_bytes() == two.read_bytes()
    df = pa.read_particlefile(two)
    assert len(df) > n_total and df["particle_id"].max() == n_total - 1


# ---- ParticleSet.execute with a collective ParticleFile: every rank keeps the same write() schedule ------------------------------
class _HostEngine:
    """Stand-in for parcels_amd.engine.Engine in the CPU suite (the product has no CPU path; this fakes only the calls
    ParticleSet.execute makes around a launch, so that its multi-rank output schedule can run under gloo without a GPU)."""

    device = None

    def __init__(self):
        self.device_variables = []
        self.data = None

    def bind_particles(self, data):
        self.data = data

    def h2d(self):
        pass

    def d2h(self, cols=None):
        pass

    def compact_deleted(self, data):
        from parcels_amd import StatusCode

        keep = data["state"] != StatusCode.Delete
        self.data = {k: v[keep] for k, v in data.items()}
        return self.data


def _host_launch(self, pset, endtime, dt, have_guess0=0):
    """Kernel.launch on the host columns: released particles move to `endtime`; particle ids divisible by 3 are deleted once they
    have reached t = 1800 (that empties the whole second shard of the test below)."""
    from parcels_amd import StatusCode

    d = pset._engine().data
    live = d["t"] <= endtime
    d["x"][live] += (endtime - d["t"][live]) * 1e-4
    d["t"][live] = endtime
    d["state"][:] = StatusCode.Evaluate
    gone = live & (endtime >= 1800.0) & ((d["particle_id"] % 3 == 0) | (d["particle_id"] >= 50))
    d["state"][gone] = StatusCode.Delete
    sc = {int(StatusCode.Evaluate): int((~gone).sum())}
    if gone.any():
        sc[int(StatusCode.Delete)] = int(gone.sum())
    return {"state_counts": sc, "steps": int(live.sum()), "kernel_ms": 0.0, "sort_ms": 0.0, "launches": 1, "attempts": 0}


def _execute_run(path, shard, n_total=100):
    import parcels_amd as pa
    from parcels_amd.kernel import Kernel

    fs = _make_fieldset()
    ids = np.arange(n_total)
    # unequal release times: the first shard starts at t = 0, the second one (ids >= 50) at t = 1200 -- and loses every particle at
    # t = 1800 (ids >= 50 and the multiples of 3 are deleted); with world size 3 the last shard would start empty as well
    t = np.where(ids < 50, 0.0, 1200.0)
    pset = pa.ParticleSet(fs, x=ids * 0.1, y=ids * 0.0, z=ids * 0.0, t=t, shard=shard)
    eng = _HostEngine()
    pset._engine = lambda: eng
    pset.async_output = False  # (the snapshot writer needs the real engine)
    old = Kernel.launch
    Kernel.launch = _host_launch
    try:
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            pset.execute([pa.AdvectionRK4], dt=600.0, runtime=3600.0, output_file=pa.ParticleFile(path, outputdt=600.0))
    finally:
        Kernel.launch = old
    return len(pset)


def _exec_worker(rank, world, port, path, q):
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _execute_run(path, "auto")))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_execute_keeps_the_collective_write_schedule_on_every_rank(tmp_path, world):
    """ParticleSet.execute with a multi-rank ParticleFile: shards with different release times, a shard that empties halfway and
    (world 3 of 100 ids released late) ranks whose particles do not exist yet all make the same write() calls -- no hang, and the
    file is the single-process file byte for byte (the start time, the output times and the stop come from ALL shards)."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    one = tmp_path / "one.parquet"
    left = _execute_run(str(one), None)
    many = tmp_path / "many.parquet"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exec_worker, args=(r, world, port, str(many), q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0, "a rank hung or failed"
    counts = dict(q.get(timeout=10) for _ in range(world))
    assert sum(counts.values()) == left and counts[world - 1] == 0  # the last shard lost every particle
    assert one.read_bytes() == many.read_bytes()


# ---- the batch-wide rules of Kernel.execute across shards (DeviceEngine.execute's passes, parcels_amd.distributed.batch_agreement) ----
class _ScriptedLib:
    """What DeviceEngine.execute calls of the library, answering from a script: one (first_error_iter, first_time_error_key) per pass."""

    def __init__(self, script):
        self.script, self.passes = list(script), []

    def _fill(self, st_ref, cap, keys):
        err, key = self.script[len(self.passes)]
        self.passes.append((int(cap), [int(k) for k in keys]))
        st = st_ref._obj
        st.first_error_iter, st.first_time_error_key, st.paused, st.launches = err, key, 0, 1
        st.steps = 100
        if err and (cap == 0 or err <= cap):
            st.state_counts[70] = 3  # (somebody ends the pass in an error state)
        return 0

    def pk_execute_begin(self, h, prm_ref):
        p = prm_ref._obj
        self._next = (p.max_iters, [p.twe_key[k] for k in range(p.twe_n)])
        return 0

    def pk_execute_end(self, h, st_ref):
        return self._fill(st_ref, *self._next)

    def pk_execute_rerun_keys(self, h, cap, n, keys, st_ref):
        return self._fill(st_ref, cap, [keys[k] for k in range(n)])


def _scripted_engine(script):
    import types

    from parcels_amd import _hip
    from parcels_amd.engine import DeviceEngine

    eng = object.__new__(DeviceEngine)
    eng.lib = _ScriptedLib(script)
    eng.ctx = types.SimpleNamespace(check=lambda rc, what=None: None, handle=None)
    eng.windowed, eng.exact_error_stop, eng.agree_min, eng.agree_codes, eng.device = False, True, None, None, None

    def make_params(kernel_ids, *, max_iters=0, twe_keys=(), **kw):
        p = _hip.ExecParams()
        import ctypes as C

        p.max_iters, p.twe_n = int(max_iters), len(twe_keys)
        if twe_keys:
            p._twe_keys = (C.c_int64 * len(twe_keys))(*[int(k) for k in twe_keys])
            p.twe_key = C.cast(p._twe_keys, C.POINTER(C.c_int64))
        return p

    eng.make_params = make_params
    return eng


def test_execute_passes_follow_the_reference_batch_rules():
    """DeviceEngine.execute's second look at a call: a sample outside the time interval in iteration 5 is listed and the call repeated
    BEFORE the error stop of the same iteration is applied (kernel.py:236-245 acts on what the iteration left behind); an error stop in an
    earlier iteration wins, and the time error behind it never happens."""
    k5 = (5 << 32) | 2003
    eng = _scripted_engine([(5, k5), (5, 0), (5, 0)])
    st = eng.execute([4], endtime=10.0, dt0=1.0)
    assert eng.lib.passes == [(0, []), (0, [k5]), (5, [k5])]
    assert st["reran"] == 2 and st["first_error_iter"] == 5 and st["time_error_keys"] == [k5]
    eng = _scripted_engine([(3, (7 << 32) | 1), (3, 0)])
    st = eng.execute([4], endtime=10.0, dt0=1.0)
    assert eng.lib.passes == [(0, []), (3, [])] and st["time_error_keys"] == [] and st["first_error_iter"] == 3
    # a later pass finds an EARLIER sample: the later key was found on a trajectory that no longer exists and is dropped
    a, b = (9 << 32) | 1, (4 << 32) | 2
    eng = _scripted_engine([(0, a), (0, b), (0, 0)])
    st = eng.execute([4], endtime=10.0, dt0=1.0)
    assert eng.lib.passes == [(0, []), (0, [a]), (0, [b])] and st["time_error_keys"] == [b]


def _agree_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from parcels_amd.distributed import batch_agreement

        k5 = (5 << 32) | 2003
        # rank 0 holds the particle that leaves the time interval (iteration 5) and errs; rank 1 steps happily; rank 2 (if any) is empty
        script = {0: [(5, k5), (5, 0), (5, 0)], 1: [(0, 0), (0, 0), (0, 0)]}.get(rank)
        eng = _scripted_engine(script or [])
        eng.agree_min, eng.agree_codes = batch_agreement()
        st = eng.execute([4], endtime=10.0, dt0=1.0) if script else eng.execute_idle()
        q.put((rank, eng.lib.passes, st["first_error_iter"], st["time_error_keys"], st["codes_any_shard"], st["reran"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shards_are_one_batch_for_the_error_stop_and_the_time_error(world):
    """world 2 / 3 over gloo: the shard WITHOUT the offending particle repeats its call with the same listed sample and the same iteration
    limit as the shard with it, an empty shard keeps the schedule of agreements, and every rank learns the error code to raise."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, "a rank hung or failed"
    res = {r[0]: r[1:] for r in (q.get(timeout=10) for _ in range(world))}
    k5 = (5 << 32) | 2003
    for rank in range(world):
        passes, cap, keys, codes, reran = res[rank]
        assert cap == 5 and keys == [k5] and codes == [70] and reran == 2, (rank, res[rank])
        if rank < 2:
            assert passes == [(0, []), (0, [k5]), (5, [k5])], (rank, passes)


# ---- every rank raises the exception of the BATCH (round-4 ADVICE: the rank without the erring particle raised TypeError) ----------
def _raise_launch_factory(codes_by_rank):
    def launch(self, pset, endtime, dt, have_guess0=0):
        from parcels_amd import StatusCode
        from parcels_amd.distributed import batch_agreement
        from parcels_amd.engine import DeviceEngine

        d = pset._engine().data
        d["state"][:] = StatusCode.Evaluate
        mine = codes_by_rank.get(dist.get_rank(), [])
        for k, code in enumerate(mine):
            d["state"][k] = code
        counts = {int(c): 1 for c in mine}
        _, agree_codes = batch_agreement()
        present = agree_codes([1 if counts.get(code) else 0 for code in DeviceEngine._RAISING_CODES])
        return {"state_counts": counts or {int(StatusCode.Evaluate): len(d["state"])}, "steps": 1, "kernel_ms": 0.0, "sort_ms": 0.0, "launches": 1,
                "attempts": 0, "codes_any_shard": [c for c, p in zip(DeviceEngine._RAISING_CODES, present) if p]}

    return launch


def _raise_worker(rank, world, port, path, codes_by_rank, q):
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import warnings

        import parcels_amd as pa
        from parcels_amd.kernel import Kernel

        fs = _make_fieldset()
        ids = np.arange(40)
        pset = pa.ParticleSet(fs, x=ids * 0.1, y=ids * 0.0, z=ids * 0.0, t=np.zeros(40), shard="auto")
        eng = _HostEngine()
        pset._engine = lambda: eng
        pset.async_output = False
        Kernel.launch = _raise_launch_factory(codes_by_rank)
        raised = None
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                pset.execute([pa.AdvectionRK4], dt=600.0, runtime=3600.0, output_file=pa.ParticleFile(path, outputdt=600.0))
        except Exception as e:  # noqa: BLE001 -- the test compares the TYPE every rank ends with
            raised = type(e).__name__
        q.put((rank, raised))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("codes_by_rank,expect", [({0: [60]}, "FieldOutOfBoundError"), ({1: [70]}, "OutsideTimeInterval"),
                                                    ({0: [60], 1: [70]}, "OutsideTimeInterval"), ({1: [52, 61]}, "FieldOutOfBoundSurfaceError")])
def test_every_rank_raises_the_exception_of_the_batch(tmp_path, codes_by_rank, expect):
    """Only one shard holds the erring particle (or the shards hold different codes): every rank must raise the SAME exception type, the
    first of ErrorsToThrow over all shards (kernel.py:31-38, 236-245)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_raise_worker, args=(r, world, port, str(tmp_path / "o.parquet"), codes_by_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, "a rank hung or failed"
    res = dict(q.get(timeout=10) for _ in range(world))
    assert res == {0: expect, 1: expect}, res


# ---- round 5: all failing samples of a pass are listed at once and VALIDATED by the next pass (DeviceEngine._twe_step) ----------------
def _k(it, s):
    return (it << 32) | s


def test_twe_step_lists_everything_found_and_validates_it():
    from parcels_amd.engine import DeviceEngine as E

    found = [_k(5, 0), _k(5, 1), _k(6, 0), _k(6, 1)]
    # pass 1: nothing listed, four samples fail for somebody -> all four listed at once
    d, keys, cap = E._twe_step([], 0, 0, found, [], speculative=True)
    assert (d, keys, cap) == ("key", found, 0)
    # pass 2: every listed key justified, nothing new -> the call stands
    assert E._twe_step(keys, 0, 0, [], [True] * 4, speculative=True) == (None, found, 0)
    # ... or: the third key was found on a trajectory that no longer exists -> dropped, the later justified one stays (speculatively)
    d, keys2, _ = E._twe_step(keys, 0, 0, [], [True, True, False, True], speculative=True)
    assert d == "key" and keys2 == [_k(5, 0), _k(5, 1), _k(6, 1)]
    # without validation (speculative=False) only the first problem is fixed and everything behind it goes: the round-4 scheme
    assert E._twe_step([], 0, 0, found, [], speculative=False) == ("key", [_k(5, 0)], 0)
    assert E._twe_step([_k(9, 1)], 0, 0, [_k(4, 2)], [True], speculative=False) == ("key", [_k(4, 2)], 0)


def test_twe_step_error_stop_and_iteration_limit():
    from parcels_amd.engine import DeviceEngine as E

    # an error in iteration 3 stops the batch before the sample of iteration 7 fails: the key never happens
    assert E._twe_step([], 0, 3, [_k(7, 1)], [], speculative=True) == ("cap", [], 3)
    # same iteration: the failing sample is listed BEFORE the error stop of that iteration is applied (kernel.py:236-245 acts on what it left)
    assert E._twe_step([], 0, 5, [_k(5, 2003)], [], speculative=True) == ("key", [_k(5, 2003)], 0)
    assert E._twe_step([_k(5, 2003)], 0, 5, [], [True], speculative=True) == ("cap", [_k(5, 2003)], 5)
    assert E._twe_step([_k(5, 2003)], 5, 5, [], [True], speculative=True) == (None, [_k(5, 2003)], 5)
    # a listed key beyond the iteration limit is never reached: not "unjustified"
    assert E._twe_step([_k(2, 0), _k(9, 0)], 4, 4, [], [True, False], speculative=True) == (None, [_k(2, 0), _k(9, 0)], 4)


class _ReportingLib(_ScriptedLib):
    """_ScriptedLib whose passes also answer pk_execute_twe_report: script entries (err, found keys, hits of the listed keys)."""

    def _fill(self, st_ref, cap, keys):
        err, found, hits = self.script[len(self.passes)]
        self.passes.append((int(cap), [int(k) for k in keys]))
        self._last = (found, hits)
        st = st_ref._obj
        st.first_error_iter, st.first_time_error_key, st.paused, st.launches = err, (min(found) if found else 0), 0, 1
        st.steps = 100
        return 0

    def pk_execute_twe_report(self, h, found, cap, nf, hit, n_listed):
        f, hits = self._last
        for i, k in enumerate(f):
            found[i] = k
        nf._obj.value = len(f)
        for i in range(n_listed):
            hit[i] = int(bool(hits[i]))
        return 0


def test_execute_needs_two_passes_for_many_keys_not_one_per_key():
    """The seed-9501 shape (profiles/r04: 105 passes): a particle past the last level fails EVERY later sample.  Pass 1 finds them all, pass 2
    runs with all of them listed and reports every one justified: done."""
    many = [_k(it, s) for it in range(20, 30) for s in range(6)]
    eng = _scripted_engine([])
    eng.lib = _ReportingLib([(0, many, []), (0, [], [True] * len(many))])
    st = eng.execute([4], endtime=10.0, dt0=1.0)
    assert st["reran"] == 1 and st["time_error_keys"] == many and eng.lib.passes == [(0, []), (0, many)]
    # a speculative key that does not survive validation costs one more pass, not one per key
    eng = _scripted_engine([])
    hits2 = [True] * len(many)
    hits2[30] = False
    rest = many[:30] + many[31:]
    eng.lib = _ReportingLib([(0, many, []), (0, [], hits2), (0, [], [True] * len(rest))])
    st = eng.execute([4], endtime=10.0, dt0=1.0)
    assert st["reran"] == 2 and st["time_error_keys"] == rest


def _agree_keys_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from parcels_amd.distributed import batch_agreement

        a, b = [_k(it, s) for it in (7, 8) for s in range(4)], [_k(it, s) for it in (8, 9) for s in range(4)]
        both = sorted(set(a) | set(b))
        # rank 0 finds the samples of iterations 7-8, rank 1 those of 8-9 (its particles cross the last level an iteration later); in the
        # validation pass rank 0 justifies its own keys only, rank 1 its own: together all of them.  Rank 2 (if any) is empty.
        script = {0: [(0, a, []), (0, [], [k in a for k in both])], 1: [(0, b, []), (0, [], [k in b for k in both])]}.get(rank)
        eng = _scripted_engine([])
        if script:
            eng.lib = _ReportingLib(script)
        eng.agree_min, eng.agree_codes = batch_agreement()
        st = eng.execute([4], endtime=10.0, dt0=1.0) if script else eng.execute_idle()
        q.put((rank, st["time_error_keys"], st["reran"], eng.lib.passes if script else None, both))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shards_agree_on_all_failing_samples_of_a_pass(world):
    """world 2 / 3 over gloo: every rank lists the UNION of the failing samples the shards found, a key only another shard justifies stands,
    and the call is over after the validation pass -- two passes, not one per key, on every rank (the empty one included)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_keys_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, "a rank hung or failed"
    res = {r[0]: r[1:] for r in (q.get(timeout=10) for _ in range(world))}
    for rank in range(world):
        keys, reran, passes, both = res[rank]
        assert keys == both and reran == 1, (rank, res[rank])
        if passes is not None:
            assert passes == [(0, []), (0, both)], (rank, passes)


class _FailingLib(_ScriptedLib):
    def pk_execute_end(self, h, st_ref):
        raise RuntimeError("field window too small: a single step does not fit into the resident time levels")


def _abort_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from parcels_amd.distributed import batch_agreement

        eng = _scripted_engine([(0, 0), (0, 0)])
        if rank == 1:
            eng.lib = _FailingLib([])
        eng.agree_min, eng.agree_codes = batch_agreement()
        try:
            eng.execute([4], endtime=10.0, dt0=1.0) if rank < 2 else eng.execute_idle()
            q.put((rank, None))
        except Exception as e:  # noqa: BLE001
            q.put((rank, type(e).__name__))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_a_rank_that_raises_inside_a_pass_takes_the_others_with_it(world):
    """Round-4 ADVICE: execute() may raise between two agreements (a ring too small for one step, too many failing samples); the other
    ranks -- the idle one included -- must not wait in the next all-reduce for a rank that is gone: they raise CollectiveAbort at the agreement
    the failing rank still attends."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_abort_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=90)
        assert p.exitcode == 0, "a rank hung or failed"
    res = dict(q.get(timeout=10) for _ in range(world))
    assert res[1] == "RuntimeError" and all(res[r] == "CollectiveAbort" for r in range(world) if r != 1), res


def _store_abort_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from parcels_amd import distributed as D

        agree_min, _ = D.batch_agreement()
        agree_min(0, 0)  # a first pass every rank attends
        if rank == 1:
            # fails OUTSIDE the launch loop (say, in its write): no agreement sits there -- it leaves the note and goes
            D.post_abort(RuntimeError("disk full"))
            q.put((rank, "RuntimeError"))
            # ... the user catches, and the next collective execute() of this rank starts by withdrawing the note
            dist.barrier()
            D.clear_abort()
            dist.barrier()
            D.check_abort()
            q.put((rank, "clean"))
            return
        import time as _t

        for _ in range(200):  # (the note travels through the rank-0 store: poll instead of assuming an order)
            try:
                D.check_abort()
            except D.CollectiveAbort as e:
                q.put((rank, "CollectiveAbort:" + str("disk full" in str(e))))
                break
            _t.sleep(0.01)
        else:
            q.put((rank, "no note"))
        # the same through the entry points that call it: the next agreement / the write gather raise instead of blocking
        for fn in (lambda: agree_min(0, 0), lambda: D.gather_rows_to_root({})):
            try:
                fn()
                q.put((rank, "entered a collective"))
            except D.CollectiveAbort:
                q.put((rank, "stopped"))
        dist.barrier()
        dist.barrier()
        D.check_abort()
        q.put((rank, "clean"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_a_rank_that_fails_outside_the_agreements_leaves_a_note(world):
    """a rank-local failure after the per-pass agreement (restore, key exchange, write, compaction) must not leave the others
    waiting in their next collective: post_abort() leaves a note in the group's store, agree_min / gather_rows_to_root read it first."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_store_abort_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=90)
        assert p.exitcode == 0, "a rank hung or failed"
    got = {}
    while not q.empty():
        r, v = q.get(timeout=10)
        got.setdefault(r, []).append(v)
    assert got[1] == ["RuntimeError", "clean"], got
    for r in range(world):
        if r != 1:
            assert got[r] == ["CollectiveAbort:True", "stopped", "stopped", "clean"], got


class _GlooComm:
    """Stand-in for the library's RCCL communicator (DeviceEngine.comm_allreduce / comm_allgather = pk_comm_allreduce_i64 / _allgather_i64) over
    gloo: what the C-ABI branch of distributed.batch_agreement calls, so that ITS logic runs at world 2 / 3 on the CPU (the entry points themselves
    are tested through ctypes on the GPU, tests/test_gpu_comm_cabi.py; RCCL refuses two ranks on one device)."""

    def __init__(self):
        self.comm = (dist.get_rank(), dist.get_world_size())
        self.calls = {"allreduce": 0, "allgather": 0}

    def comm_init(self, *a):  # (ensure_comm looks for the method)
        raise AssertionError("not reached: ensure_comm is patched")

    def comm_allreduce(self, values, op):
        import torch

        t = torch.tensor([int(v) for v in values], dtype=torch.int64)
        dist.all_reduce(t, op={"min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}[op])
        self.calls["allreduce"] += 1
        return t.numpy()

    def comm_allgather(self, values):
        import torch

        t = torch.tensor([int(v) for v in values], dtype=torch.int64)
        parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, t)
        self.calls["allgather"] += 1
        return np.stack([p.numpy() for p in parts])


def _agree_keys_cabi_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import parcels_amd.distributed as D

        comm = _GlooComm()
        D.ensure_comm = lambda engine, group=None: engine is comm  # the C-ABI branch for THIS engine
        a, b = [_k(it, s) for it in (7, 8) for s in range(4)], [_k(it, s) for it in (8, 9) for s in range(4)]
        both = sorted(set(a) | set(b))
        script = {0: [(0, a, []), (0, [], [k in a for k in both])], 1: [(0, b, []), (0, [], [k in b for k in both])]}.get(rank)
        eng = _scripted_engine([])
        if script:
            eng.lib = _ReportingLib(script)
        eng.agree_min, eng.agree_codes = D.batch_agreement(engine=comm)
        assert eng.agree_min.transport == "c-abi"
        st = eng.execute([4], endtime=10.0, dt0=1.0) if script else eng.execute_idle()
        codes = eng.agree_codes([rank == 1, False, False])
        q.put((rank, st["time_error_keys"], st["reran"], eng.lib.passes if script else None, both, codes, dict(comm.calls)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_the_c_abi_branch_of_the_agreements_at_world_2_and_3(world):
    """Round 6: under RCCL the agreements go through the library's own communicator (pk_comm_allreduce_i64 / pk_comm_allgather_i64).  The same
    scenario as above through THAT branch of batch_agreement, the communicator stood in for by gloo: union of the keys, OR of the justifications,
    error codes present on any rank, two passes on every rank -- and no torch collective of the old path is used for them."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_keys_cabi_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, "a rank hung or failed"
    res = {r[0]: r[1:] for r in (q.get(timeout=10) for _ in range(world))}
    for rank in range(world):
        keys, reran, passes, both, codes, calls = res[rank]
        assert keys == both and reran == 1, (rank, res[rank])
        assert codes == [1, 0, 0], (rank, codes)
        assert calls["allgather"] >= 1 and calls["allreduce"] >= 3, (rank, calls)
        if passes is not None:
            assert passes == [(0, []), (0, both)], (rank, passes)
