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

    def counting(group=None, device=None):
        amin, acodes = real(group, device)

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
