"""oracle/fast_agrid_cpu.c (the CPU-idiomatic restatement of the headline workload that bench.py times as `cpu_baseline`) against
oracle/parcels_oracle.c (pinned to the reference at rtol 0 by tests/test_oracle_golden.py): bit for bit, on spherical and flat
meshes, a last step clipped to the end time, cell-sorted or not, one thread or several."""

import numpy as np
import pytest

from case_utils import run_oracle
from oracle import c_oracle as co
from oracle import cases


@pytest.mark.parametrize("mesh", ["spherical", "flat"])
@pytest.mark.parametrize("dt", [3600.0, 1000.0])
def test_fast_cpu_port_equals_the_oracle_bit_for_bit(mesh, dt):
    case = cases.rect_agrid_case("fastcpu", mesh=mesh, kernels=["AdvectionRK4"], seed=4, nx=40, ny=24, nz=7, nt=5, npart=4000, dt=dt, runtime=26 * 3600.0 + 777.0)
    ref, err, st = run_oracle(case)
    assert err is None
    for sort, threads in ((True, 1), (False, 3), (True, 4)):
        got, steps, _ = co.fast_rk4_agrid(case, endtime=case["runtime"], nthreads=threads, sort_by_cell=sort)
        assert steps == st["steps"]
        for k in ("x", "y", "t"):
            assert np.array_equal(got[k], ref[k]), (k, sort, threads)
        assert np.all(got["state"] == 1) and np.all(ref["state"] == 1)


def test_fast_cpu_port_refuses_what_it_does_not_cover():
    case = cases.rect_agrid_case("fastcpu_escape", mesh="flat", kernels=["AdvectionRK4"], seed=12, vel=6.0, margin=0.01, dt=1800.0, runtime=20 * 1800.0)
    with pytest.raises(ValueError, match="leave the domain"):
        co.fast_rk4_agrid(case, endtime=case["runtime"])


def test_the_bench_workload_itself():
    """bench.py's C2 FieldSet at a reduced particle count: the port reproduces the oracle there too."""
    from bench import c2_case

    case = c2_case(seed=1, lo=0, hi=3000, nx=90, ny=45, nz=12, nt=4)
    case["runtime"] = 12 * 3600.0
    ref, err, st = run_oracle(case)
    got, steps, _ = co.fast_rk4_agrid(case, endtime=case["runtime"], nthreads=2)
    assert err is None and steps == st["steps"] == 3000 * 12
    for k in ("x", "y", "t"):
        assert np.array_equal(got[k], ref[k]), k
