#!/usr/bin/env python
"""TEST / MEASUREMENT INFRASTRUCTURE (build container only: needs /root/reference).

Times the REFERENCE ITSELF -- Parcels' own ParticleSet.execute(AdvectionRK4), unmodified, under oracle/ref_shim.py -- on the
C2 workload of bench.py (same FieldSet, a subset of the same particles, 24 RK4 steps), SURVEY.md section 8d(1)/(2):
  * one process (the reference is single-threaded);
  * NPROC processes, particles sharded by id -- the only multi-core mode the reference admits.
Writes profiles/<out>.json, which bench.py attaches to its line as `cpu_baseline_reference` (the GPU box has no
/root/reference, so this number cannot be re-measured there; box and core count are recorded with it).

    python tools/time_reference_cpu.py --particles 200000 --procs 8 --out r02_cpu_reference
"""
import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run_shard(args):
    lo, hi, steps = args
    import warnings

    import numpy as np

    warnings.simplefilter("ignore")
    from bench import c2_case
    from oracle import make_golden as mg

    case = c2_case(seed=1, lo=lo, hi=hi)  # particles lo..hi-1 of bench.py's id space
    case["runtime"] = steps * case["dt"]
    t0 = time.perf_counter()
    out, err, _ = mg.ref_run_case(case)
    el = time.perf_counter() - t0
    assert err is None and np.all(out["t"] == case["runtime"]), (err,)
    return (hi - lo) * steps, el


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=200_000)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default="cpu_reference")
    a = ap.parse_args()
    n, K = a.particles, a.steps
    # (1) single process
    t0 = time.perf_counter()
    work, el_inner = _run_shard((0, n // a.procs, K))
    single = {"particles": n // a.procs, "steps": K, "seconds": el_inner, "value": work / el_inner}
    # (2) all cores, sharded by id
    edges = [n * k // a.procs for k in range(a.procs + 1)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(a.procs) as pool:
        res = pool.map(_run_shard, [(edges[k], edges[k + 1], K) for k in range(a.procs)])
    wall = time.perf_counter() - t0
    inner = max(r[1] for r in res)
    out = {
        "kind": "reference",
        "what": "Parcels v4-alpha ParticleSet.execute(AdvectionRK4) under oracle/ref_shim.py, C2 FieldSet of bench.py (fp64, 360x180x50x24)",
        "unit": "particle-steps/s",
        "single_process": single,
        "all_cores": {"processes": a.procs, "particles": n, "steps": K, "seconds_slowest_shard": inner, "seconds_wall_incl_startup": wall,
                      "value": n * K / inner},
        "box": {"cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t"),
                "logical_cores": os.cpu_count(), "where": "build container (no GPU); /root/reference is not present on the GPU box"},
        "command": " ".join(sys.argv),
    }
    path = os.path.join(ROOT, "profiles", a.out + ".json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
