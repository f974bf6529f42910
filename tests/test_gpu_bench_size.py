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
