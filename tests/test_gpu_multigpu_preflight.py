"""GPU: the multi-GPU path of the product on ONE GPU, backend "nccl" (= RCCL), world size 1 -- everything a rank does in an N-rank run
except having neighbours: ParticleSet.execute with a collective ParticleFile (lock-step interval schedule, device write filter,
gather of the surviving device rows to rank 0 over RCCL, one table per output time) and the batch agreements of Kernel.execute (the
int64 all-reduces of parcels_amd.distributed.batch_agreement inside DeviceEngine.execute's passes).  The N > 1 forms of the same code
run over gloo in tests/test_distributed_cpu.py; the 8-GPU run itself is the driver's."""

import os
import socket

import numpy as np
import pytest

from case_utils import build_fieldset, build_pset, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture
def rccl_world1():
    import torch
    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def _run(case, path, distributed, kernels):
    import warnings

    import parcels_amd as pa

    fs = build_fieldset(case)
    pset = build_pset(case, fs, sort_by_cell=True)
    pf = pa.ParticleFile(path, outputdt=2 * abs(case["dt"]), distributed=distributed)
    err = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            pset.execute(kernels, dt=float(case["dt"]), runtime=float(case["runtime"]), output_file=pf)
        except (pa.OutsideTimeInterval, pa.FieldOutOfBoundError) as e:
            err = type(e).__name__
    return pset, pf, err


def test_collective_write_out_of_the_product_over_rccl(gpu, rccl_world1, tmp_path):
    """ParticleFile(distributed="always") in a 1-rank RCCL group: ParticleSet.execute -> ParticleFile.write -> device_write_rows ->
    gather_write_columns (device tensors) -> rank 0's table.  Byte for byte the file of the rank-local path; with deletions on the way."""
    import parcels_amd as pa

    case, _, _ = load_golden("agrid_flat_rk4_3d_escape_delete")
    kernels = [pa.AdvectionRK4_3D, pa.kernels.DeleteParticle]
    a, pfa, ea = _run(case, str(tmp_path / "local.parquet"), False, kernels)
    b, pfb, eb = _run(case, str(tmp_path / "collective.parquet"), "always", kernels)
    assert ea is None and eb is None
    assert pfb._collective and pfb.gather_seconds > 0.0  # the exchange ran
    assert len(a) == len(b) and 0 < len(a) < len(case["x"])
    for k in ("x", "y", "z", "t", "particle_id"):
        assert np.array_equal(a._data[k], b._data[k]), k
    assert (tmp_path / "local.parquet").read_bytes() == (tmp_path / "collective.parquet").read_bytes()


@pytest.mark.parametrize("name", ["twe_agrid_sph_rk4_raise", "agrid_flat_rk4_escape_stagger"])
def test_batch_agreements_run_over_rccl(gpu, rccl_world1, tmp_path, name):
    """The error stop and the call-wide time error of a collective run go through batch_agreement's all-reduces (here: a group of one,
    so the agreed values are the rank's own): same exception and same columns as the rank-local run with the same output schedule
    (which the reference fixtures pin, tests/test_gpu_parity.py)."""
    import parcels_amd as pa
    import parcels_amd.distributed as pd

    case, _, err = load_golden(name)
    calls = {"min": 0, "codes": 0}
    real = pd.batch_agreement

    def counting(group=None, device=None, engine=None):
        amin, acodes = real(group, device, engine=engine)
        assert amin.transport == "c-abi"  # (the all-reduces run through pk_comm_allreduce_i64: RCCL inside the library)

        def m(e, k):
            calls["min"] += 1
            return amin(e, k)

        def c(p):
            calls["codes"] += 1
            return acodes(p)

        return m, c

    local, _, lerr = _run(case, str(tmp_path / "local.parquet"), False, [pa.AdvectionRK4])
    pd.batch_agreement = counting
    try:
        coll, pf, cerr = _run(case, str(tmp_path / "collective.parquet"), "always", [pa.AdvectionRK4])
    finally:
        pd.batch_agreement = real
    assert lerr == cerr == err and err is not None
    assert calls["min"] >= 2 and calls["codes"] >= 1  # at least one repeated pass, and the final agreement on the codes
    for k in local._data:
        assert np.array_equal(local._data[k], coll._data[k], equal_nan=True), k
    assert (tmp_path / "local.parquet").read_bytes() == (tmp_path / "collective.parquet").read_bytes()


@pytest.mark.parametrize("name", ["twe_agrid_sph_rk4_raise", "twe_agrid_sph_ee_delete"])
def test_two_shards_are_one_batch_on_the_real_kernels(gpu, tmp_path, name):
    """Two ranks (sharing this box's one GPU, gloo) run ONE id space sharded by id through the real kernels.  The particles are ordered
    so that the LATE releases -- the ones that leave the field's time interval first -- all live on shard 0: in the iteration in which
    they do, the particles of shard 1 are still inside it.  The reference fails that sample for every particle of the call
    (field.py:31-44), so shard 1 must repeat its launch with the sample listed although none of ITS particles ever left the interval
    (parcels_amd.distributed.batch_agreement) -- the concatenated shards equal the reference fixture, rank 1 raises the same exception."""
    import subprocess
    import sys

    from case_utils import ROOT_DIR, compare, tolerance_for

    case, out, err = load_golden(name)
    t0 = np.asarray(case["t0"], dtype=np.float64)
    order = np.argsort(-t0, kind="stable")
    np.save(tmp_path / "order.npy", order)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT_DIR, "tests", "_twe_two_rank_worker.py"), name, str(tmp_path)]
    r = subprocess.run(cmd, cwd=ROOT_DIR, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    ranks = [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]
    for k in range(2):
        assert str(ranks[k]["err"]) == (err or ""), (k, str(ranks[k]["err"]), err)
        assert int(ranks[k]["reran"]) >= 1 and len(ranks[k]["keys"]) >= 1  # BOTH shards repeated the call with the listed sample
    assert np.array_equal(ranks[0]["keys"], ranks[1]["keys"])
    got = {k: np.concatenate([ranks[0][k], ranks[1][k]]) for k in ("x", "y", "z", "t", "dx", "dy", "dz", "dt", "state", "ei", "particle_id")}
    if len(out["x"]):  # the fixture keeps its particles: row i of the sharded run is input row order[i]
        ref = {k: np.asarray(out[k])[order] for k in out if k != "particle_id"}
        ref["particle_id"] = got["particle_id"]
        compare(got, ref, rtol=tolerance_for(name, case), check_state="all", label=name)
    else:  # the reference deleted every particle of the call -- per particle, shard 1 would have kept its own
        assert len(got["x"]) == 0
