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
