"""Parity of the measured configurations themselves (BASELINE configs 3 and 5 as tools/bench_configs.py builds them): a subset
of the particles of a run is re-run through the CPU oracle on the same arrays and must reproduce -- deleted set, state, ei and t
exactly, positions to 1e-12 (1e-11 for the stochastic kernel).  Here at a size that takes seconds; the full-size runs use the
same check (`tools/bench_configs.py --check 1e5`, results under profiles/)."""

from __future__ import annotations

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config", ["c3", "c5"])
def test_bench_config_subset_reproduces_through_the_oracle(gpu, config):
    from tools import bench_configs as bc

    seen = []
    # 518 x 367 x 20 curvilinear C-grid, 4 daily levels through a ring of 3, 6-hour steps so that cells are crossed and the ring turns
    res = bc.run_config(config, scale=0.12, particles=2e5, steps=11, nt=4, nslots=3, nz=20, check=20_000, emit=seen.append, dt=6 * 3600.0)
    assert len(res) == (1 if config == "c3" else 2)
    for r in res:
        c = r["check"]
        assert c["survivors"] > 0.9 * c["n_check"] and r["particle_steps"] > 0
        assert max(c["max_abs_diff"].values()) <= 1e-11 * 6000  # z in metres is the largest coordinate


def test_c4_two_ranks_write_the_single_process_file(gpu, tmp_path):
    """BASELINE config 4 end to end on ONE GPU (rehearsal: two ranks share cuda:0 and talk over gloo, because RCCL refuses two
    ranks on one device): one id space sharded by id, fields from one shared memory-mapped copy, ParticleSet.execute with a
    ParticleFile whose rows come from the all-gather of the to-write columns -- and the file equals, byte for byte, the one a
    single process writes for the whole id space (deleted particles, cell sort and ring included)."""
    import json
    import os
    import socket
    import subprocess
    import sys

    from case_utils import ROOT_DIR

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PARCELS_AMD_BENCH_REHEARSAL="1", PK_C4_DIR=str(tmp_path / "shared"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(ROOT_DIR, "tools", "bench_configs.py"), "--config", "c4", "--scale", "0.1", "--particles", "40000", "--steps", "11",
           "--nz", "12", "--dt", "21600", "--output-every", "3", "--verify-single"]
    r = subprocess.run(cmd, env=env, cwd=ROOT_DIR, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["particles_total"] == 80000 and out["byte_identical_to_single_process_file"] is True
    assert out["parquet_rows"] >= 4 * out["remaining_particles"]


@pytest.mark.gpu
def test_bench_py_runs_under_torchrun_with_two_ranks(gpu):
    """The driver's multi-GPU entry point, `python -m torch.distributed.run ... bench.py --gpus N`, rehearsed with two ranks sharing
    this box's one GPU over gloo (PARCELS_AMD_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device): one JSON line from rank 0,
    whole-job value over one id space sharded by id, the write-out all-gather of the device columns timed."""
    import json
    import os
    import socket
    import subprocess
    import sys

    from case_utils import ROOT_DIR

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PARCELS_AMD_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(ROOT_DIR, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--particles", "200000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT_DIR, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["config"]["all_states_endofloop"] is True and out["config"]["rehearsal_shared_gpu_gloo"] is True
    assert out["writeout_allgather_ms"] is not None and out["writeout_allgather_ms"] > 0
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * out["steps"] - 2 * 200000 * 4) < 1e-6 * 2 * 200000 * 4  # value x time = all ranks' steps
    # round 4: the timed region is repeated (median quoted), both write-out collectives are timed, the line says what it ran on
    tr = out["timed_reps"]
    assert tr["n"] == 7 and tr["statistic"] == "median" and tr["wall_ms"]["min"] <= tr["wall_ms"]["median"] <= tr["wall_ms"]["max"]
    assert abs(tr["wall_ms"]["median"] - out["ms_per_step"] * out["steps"]) < 1e-9 * tr["wall_ms"]["median"] + 1e-12
    assert len(tr["kernel_ms_slowest_vs_fastest_rank_of_median_rep"]) == 2
    assert out["writeout_gather_to_root_ms"] is not None and out["writeout_gather_to_root_ms"] > 0
    assert out["comm"]["n_ranks_seen"] == 2 and out["comm"]["backend"] == "gloo" and out["device"]["compute_units"] > 0


@pytest.mark.gpu
def test_bench_py_starts_its_own_ranks(gpu):
    """`python bench.py --gpus 2` WITHOUT a launcher: bench.py re-executes itself under torch.distributed.run (one rank per GPU; on
    this 1-GPU box the ranks share cuda:0 over gloo and the line says so) and still prints exactly one JSON line with n_gpus = 2."""
    import json
    import os
    import subprocess
    import sys

    from case_utils import ROOT_DIR

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PARCELS_AMD_BENCH_REHEARSAL")}
    cmd = [sys.executable, os.path.join(ROOT_DIR, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--particles", "200000", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT_DIR, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["all_states_endofloop"] is True
    import torch

    assert out["config"].get("rehearsal_shared_gpu_gloo", False) == (torch.cuda.device_count() < 2)


@pytest.mark.gpu
def test_bench_py_secondary_configs_carry_the_oracle_check(gpu):
    """The bench line's `secondary` array (BASELINE configs 3 and 5 next to the C2 headline), here on a shrunken grid: one entry per
    kernel list with steps/s, the algorithmic-byte roofline fraction and the verdict of the oracle re-run of a subset."""
    import json
    import os
    import subprocess
    import sys

    from case_utils import ROOT_DIR

    cmd = [sys.executable, os.path.join(ROOT_DIR, "bench.py"), "--steps", "4", "--warmup", "1", "--particles", "200000", "--no-cpu-baseline",
           "--secondary-scale", "0.1", "--secondary-particles", "100000", "--secondary-check", "5000"]
    r = subprocess.run(cmd, cwd=ROOT_DIR, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    sec = out["secondary"]
    assert [e["kernels"] for e in sec] == ["AdvectionRK4_3D", "AdvectionRK45", "AdvectionDiffusionM1"], sec
    for e in sec:
        assert e["check"]["passed"] is True and e["check"]["n_check"] == 5000 and e["value"] > 0
        assert e["roofline"]["bound"] == "hbm" and 0 < e["roofline"]["frac"] < 1
        ks = e["kernel_ms_stats"]  # one cold launch, then the median of the timed ones
        assert ks["n"] == 5 and ks["min"] <= ks["median"] <= ks["max"] and e["kernel_ms"] == ks["median"] and ks["cold"] > 0
    lr = out["long_run"]
    assert lr["particle_steps"] > 0.9 * 200000 * 552 and lr["value"] > 0 and "error" not in lr
    assert out["timed_reps"]["n"] == 7 and set(out["legs_wall_s"]) >= {"secondary", "long_run", "total_after_imports"}
