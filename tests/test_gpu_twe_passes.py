"""The call-wide OutsideTimeInterval in a bounded number of passes.  Round 4 found ONE failing sample per pass over the
call: a run that overshoots the last time level with AdvectionRK45 (which overwrites the code and goes on with zeros, _advection.py:146)
failed ~100 samples and needed ~100 passes (fuzz seed 9501: 105).  Now a pass reports every failing sample (pk_execute_twe_report), they are
listed together and the next pass validates the listing -- bit for bit the same trajectories, a handful of passes."""

from __future__ import annotations

import numpy as np
import pytest

from case_utils import compare, run_hip, run_oracle, tolerance_for

pytestmark = pytest.mark.gpu


def _seed_9501(tiles=1):
    import test_gpu_fuzz as fz

    case, _ = fz.draw_case(9501)
    if tiles > 1:  # the same 662 particles over and over: the batch fails exactly the samples the small one fails
        for k in ("x", "y", "z"):
            case[k] = np.tile(np.asarray(case[k]), tiles)
    return case


def _one_key_per_pass():
    """Context manager: the round-4 scheme (no speculation), for the bit-for-bit comparison."""
    import contextlib

    from parcels_amd.engine import DeviceEngine

    @contextlib.contextmanager
    def cm():
        old = DeviceEngine.TWE_SPECULATIVE_PASSES
        DeviceEngine.TWE_SPECULATIVE_PASSES = 0
        try:
            yield
        finally:
            DeviceEngine.TWE_SPECULATIVE_PASSES = old

    return cm()


def test_seed_9501_same_bits_in_few_passes(gpu):
    case = _seed_9501()
    new, nerr, nst = run_hip(case)
    with _one_key_per_pass():
        old, oerr, ost = run_hip(case)
    assert nerr == oerr
    compare(new, old, rtol=0.0, check_state="all", label="all keys per pass vs one key per pass", skip=())
    assert nst["time_error_keys"] == ost["time_error_keys"] and len(nst["time_error_keys"]) > 50
    assert ost["reran"] >= len(ost["time_error_keys"]), "the one-key scheme should need a pass per key here"
    assert nst["reran"] <= 6, nst["reran"]
    ref, rerr, _ = run_oracle(case)
    assert rerr == nerr
    compare(new, ref, rtol=tolerance_for("fuzz9501", case), check_state="all", label="seed 9501 vs oracle")


def test_seed_9501_shape_at_1e6_particles_passes_bounded(gpu):
    tiles = 1511  # 662 x 1511 = 1 000 282 particles
    case = _seed_9501(tiles)
    got, err, st = run_hip(case)
    assert st["reran"] + 1 <= 4, f"{st['reran'] + 1} passes over the call"
    small, serr, sst = run_hip(_seed_9501())
    assert err == serr and st["time_error_keys"] == sst["time_error_keys"]
    n = len(small["x"])
    ids = got["particle_id"]
    for tile in (0, 1, tiles // 2, tiles - 1):  # every tile is the small batch again
        sel = (ids >= tile * 662) & (ids < (tile + 1) * 662)
        sub = {k: v[sel] for k, v in got.items()}
        sub["particle_id"] = sub["particle_id"] - tile * 662
        assert len(sub["x"]) == n
        compare(sub, small, rtol=0.0, check_state="all", label=f"tile {tile} of the 1e6-particle batch vs the 662-particle batch", skip=())


@pytest.mark.parametrize("name", ["twe_agrid_sph_rk45", "twe_cgrid_curv_sph_rk45_delete", "twe_agrid_sph_rk4_raise"])
def test_twe_fixtures_same_bits_both_schemes(gpu, name):
    from case_utils import load_golden

    case, out, err = load_golden(name)
    new, nerr, nst = run_hip(case)
    with _one_key_per_pass():
        old, oerr, ost = run_hip(case)
    assert nerr == oerr == err
    compare(new, old, rtol=0.0, check_state="all", label=name, skip=())
    assert nst["time_error_keys"] == ost["time_error_keys"] and nst["reran"] <= ost["reran"]


def test_overshoot_through_a_level_ring_equals_the_resident_run(gpu):
    """Many failing samples in a call of SEVERAL launches (4 levels through a ring of 3: lanes pause at the window edge, the call restarts from
    the device checkpoint for every pass): the streamed run, the resident run and the oracle agree, and the passes stay bounded."""
    from oracle import cases

    case = cases.rect_agrid_case("twe_ring_rk45", mesh="spherical", kernels=["AdvectionRK45", "DeleteOutOfBounds"], seed=141, nt=4, npart=600, stagger=True)
    case["context"] = {"RK45_tol": 500.0, "RK45_min_dt": 10.0, "RK45_max_dt": 7200.0}
    tl = float(case["time_s"][-1] - case["time_s"][0])
    case["runtime"] = tl + 3 * 3600.0 - float(np.min(case["t0"]))
    resident, rerr, rst = run_hip(case)
    ring, gerr, gst = run_hip(case, nslots=3)
    assert rerr == gerr
    assert len(rst["time_error_keys"]) > 10 and gst["time_error_keys"] == rst["time_error_keys"]
    assert gst["launches"] >= 1 and gst["reran"] <= 8 and rst["reran"] <= 6, (gst["reran"], rst["reran"])
    compare(ring, resident, rtol=0.0, check_state="all", label="ring of 3 vs resident levels", skip=())
    ref, oerr, _ = run_oracle(case)
    assert oerr == rerr
    compare(resident, ref, rtol=tolerance_for("twe_ring_rk45", case), check_state="all", label="vs oracle")
