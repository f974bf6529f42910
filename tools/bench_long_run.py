#!/usr/bin/env python
"""How fast does the gather locality of a cell-sorted particle set decay, and what re-sort cadence pays?  BASELINE config 2 (C2,
bench.py's FieldSet and particles) run over ALL its time levels -- 23 days = 552 RK4 steps of 1 h, levels resident -- unsorted,
sorted once, and re-sorted every K steps (ParticleSet(resort_every=K dt): soft time horizon + re-sort + relaunch inside one
Kernel.execute).  Prints one JSON object; profiles/r03_c2_long_run.json is its output on MI355X."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=float, default=1e7)
    ap.add_argument("--steps", type=int, default=552)
    ap.add_argument("--cadences", default="unsorted,0,192,96,48,24")
    a = ap.parse_args()
    import torch

    import parcels_amd as pa
    from bench import c2_case
    from case_utils import build_fieldset, build_pset

    case = c2_case(seed=1, lo=0, hi=int(a.particles))
    fs = build_fieldset(case)
    fs.to_device()
    eng = fs._engine
    dt = case["dt"]
    rows = []
    ref = None
    for cad in a.cadences.split(","):
        sort = cad != "unsorted"
        every = None if not sort else (0 if cad == "0" else float(cad) * dt)
        pset = build_pset(case, fs, sort_by_cell=sort, resort_every=every)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pset.execute([pa.AdvectionRK4, pa.DeleteParticle], dt=dt, runtime=a.steps * dt)  # (a few particles reach the edge of the domain in 23 days)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        st = pset._last_stats
        row = {"cadence_steps": cad, "launches": st["launches"], "kernel_ms": st["kernel_ms"], "sort_ms": st["sort_ms"],
               "particle_steps": st["steps"], "steps_per_s_kernels_and_sorts": st["steps"] / ((st["kernel_ms"] + st["sort_ms"]) * 1e-3),
               "steps_per_s_kernels": st["steps"] / (st["kernel_ms"] * 1e-3), "wall_s_incl_h2d_d2h": wall}
        row["remaining_particles"] = len(pset)
        if ref is None:
            ref = {k: np.array(pset._data[k]) for k in ("x", "y", "z", "t")}
        else:
            row["bit_identical_to_first_run"] = all(np.array_equal(ref[k], pset._data[k]) for k in ref)
        rows.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps({"workload": f"C2 360x180x50x24 fp64, {int(a.particles)} fp64 particles, AdvectionRK4, {a.steps} steps of {dt} s, levels resident",
                      "runs": rows}))


if __name__ == "__main__":
    main()
