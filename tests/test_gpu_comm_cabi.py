"""The multi-GPU exchange through the C ABI (include/parcels_hip.h: pk_comm_*, pk_gather_rows_to_root, pk_allgather_output, pk_gathered_fetch;
SURVEY.md section 8b / 8e): RCCL opened by the library itself, no torch in the path.  One GPU box: a communicator of ONE rank -- the device
write filter, the packing in host row order, the count exchange and the fetch are the code every rank runs at N > 1; the N > 1 LOGIC
(ragged counts, rank order, empty shards) is covered by the gloo tests of tests/test_distributed_cpu.py through the same Python callers."""

from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

from case_utils import build_fieldset, build_pset

pytestmark = pytest.mark.gpu


def _case(npart=20000):
    from oracle import cases

    return cases.rect_agrid_case("comm_cabi", mesh="spherical", kernels=["AdvectionRK4"], seed=21, npart=npart, runtime=None)


@pytest.fixture()
def engine_with_comm(gpu):
    import parcels_amd as pa
    from parcels_amd import _hip

    case = _case()
    fs = build_fieldset(case)
    fs.to_device(0)
    eng = fs._engine
    lib = eng.lib
    uid = (C.c_uint8 * _hip.PK_COMM_ID_BYTES)()
    assert lib.pk_comm_unique_id(uid) == 0, lib.pk_last_error(None)  # (the raw entry point; DeviceEngine.comm_unique_id wraps it)
    eng.comm_init(0, 1, bytes(uid))
    assert lib.pk_comm_init(eng.ctx.handle, 0, 1, uid) != 0  # one communicator per context
    yield pa, case, fs, eng
    eng.comm_destroy()


def test_comm_info_and_allreduce(engine_with_comm):
    from parcels_amd import _hip

    _, _, _, eng = engine_with_comm
    r, w, v = C.c_int32(-1), C.c_int32(-1), C.c_int32(0)
    eng.ctx.check(eng.lib.pk_comm_info(eng.ctx.handle, C.byref(r), C.byref(w), C.byref(v)), "pk_comm_info")
    assert (r.value, w.value) == (0, 1) and v.value > 20000  # (RCCL reports NCCL's version code: major * 10000 + ...)
    vals = np.array([7, -3, 1 << 40], dtype=np.int64)
    for op in (_hip.PK_OP_MIN, _hip.PK_OP_MAX, _hip.PK_OP_SUM):
        got = vals.copy()
        eng.ctx.check(eng.lib.pk_comm_allreduce_i64(eng.ctx.handle, got.ctypes.data_as(C.c_void_p), len(got), op), "pk_comm_allreduce_i64")
        assert np.array_equal(got, vals)
    assert eng.lib.pk_comm_allreduce_i64(eng.ctx.handle, vals.ctypes.data_as(C.c_void_p), 3, 99) != 0
    assert np.array_equal(eng.comm_allgather(vals), vals[None, :])  # pk_comm_allgather_i64: (world, n) in rank order


@pytest.mark.parametrize("sort", [True, False])
def test_gather_rows_equals_the_reference_filter_on_the_host_columns(engine_with_comm, sort):
    """Staggered release times: at an output time only the particles within dt/2 of it are written (particlefile.py:198-221).  The rows the
    library selects on the device (cell-sorted or not) and hands to rank 0 are, bit for bit and in host row order, the rows NumPy selects."""
    from parcels_amd import _hip
    from parcels_amd.particlefile import _to_write_particles

    pa, case, fs, eng = engine_with_comm
    n = len(case["x"])
    c = dict(case)
    dt = float(case["dt"])
    c["t0"] = (np.arange(n) % 4) * 2 * dt  # releases at 0, 2, 4, 6 h: at the output time 3 h half of the particles are not released yet
    pset = build_pset(c, fs, sort_by_cell=sort)
    pset.execute([pa.AdvectionRK4], dt=dt, runtime=3 * dt)
    t_out = 3 * dt
    mask = _hip.PK_COL_T | _hip.PK_COL_Z | _hip.PK_COL_Y | _hip.PK_COL_X | _hip.PK_COL_PARTICLE_ID
    for apply, to_all in ((1, False), (0, False), (1, True)):
        counts = np.zeros(1, np.int64)
        fn = eng.lib.pk_allgather_output if to_all else eng.lib.pk_gather_rows_to_root
        eng.ctx.check(fn(eng.ctx.handle, t_out, apply, mask, counts.ctypes.data_as(C.c_void_p)), "gather")
        m = int(counts[0])
        out = {"t": np.empty(m), "z": np.empty(m), "y": np.empty(m), "x": np.empty(m), "particle_id": np.empty(m, np.int64)}
        d = _hip.ParticlesDesc()
        d.n = m
        for k, a in out.items():
            setattr(d, k, a.ctypes.data_as(C.c_void_p))
        eng.ctx.check(eng.lib.pk_gathered_fetch(eng.ctx.handle, C.byref(d), m), "pk_gathered_fetch")
        host = {k: np.array(pset._data[k]) for k in ("t", "dt", "z", "y", "x", "particle_id")}
        idx = _to_write_particles(host, t_out) if apply else np.arange(n)
        assert 0 < len(idx) and (len(idx) < n) == bool(apply) and m == len(idx)
        for k in out:
            assert np.array_equal(out[k], host[k][idx]), k
    # too short a destination is an error, not an overrun
    d.n = 1
    assert eng.lib.pk_gathered_fetch(eng.ctx.handle, C.byref(d), 1) != 0


def test_particlefile_collective_path_runs_on_the_c_abi_exchange(engine_with_comm, tmp_path):
    """ParticleFile(distributed='always') in a 1-rank nccl group: DeviceEngine.gather_rows (pk_gather_rows_to_root) feeds the table, and the
    file equals the rank-local one."""
    import os
    import socket

    import torch
    import torch.distributed as dist

    pa, case, fs, eng = engine_with_comm
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dt = float(case["dt"])
        files = {}
        for mode in ("local", "collective"):
            pset = build_pset(case, fs, sort_by_cell=True)
            pf = pa.ParticleFile(tmp_path / f"{mode}.parquet", outputdt=2 * dt, distributed=("always" if mode == "collective" else False))
            before = eng.comm_stats["gathers"]
            pset.execute([pa.AdvectionRK4], dt=dt, runtime=6 * dt, output_file=pf)
            files[mode] = pa.read_particlefile(tmp_path / f"{mode}.parquet")
            if mode == "collective":
                assert eng.comm_stats["gathers"] > before, "the collective write did not go through pk_gather_rows_to_root"
        assert files["local"].equals(files["collective"])
    finally:
        dist.destroy_process_group()
