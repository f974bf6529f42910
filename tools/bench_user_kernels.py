#!/usr/bin/env python
"""What a user-written kernel costs: BASELINE config 2 (360 x 180 x 50 x 24 A-grid, fp64, 1e7 particles, AdvectionRK4, 24 steps) with the
two kernels every Parcels tutorial adds -- an ageing kernel and a delete-when-old kernel -- (a) compiled into the fused launch
(parcels_amd/jit.py), (b) on the host path (PARCELS_AMD_JIT=0: the reference's loop on the host columns, the built-in kernel's body on the
GPU), next to (c) AdvectionRK4 alone.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from parcels_amd import StatusCode  # noqa: E402


def Age(particles, fieldset):
    particles.age += particles.dt


def DeleteOld(particles, fieldset):
    particles.state = np.where(particles.age > fieldset.max_age, StatusCode.Delete, particles.state)


def SampleT(particles, fieldset):
    particles.temp = fieldset.T[particles]


def run(kernels, jit, n, steps):
    import parcels_amd as pa
    try:
        from __main__ import c2_case  # (called from bench.py itself)
    except ImportError:
        from bench import c2_case
    from case_utils import build_fieldset

    os.environ["PARCELS_AMD_JIT"] = "1" if jit else "0"
    case = c2_case(seed=1, lo=0, hi=n)
    case["fields"] = dict(case["fields"])
    case["field_dims"] = dict(case["field_dims"])
    case["fields"]["T"] = np.asarray(case["fields"]["U"]) * 3.0 + 10.0  # a tracer on the velocity grid
    case["field_dims"]["T"] = case["field_dims"]["U"]
    fs = build_fieldset(case)
    fs.add_context("max_age", 20 * 3600.0)
    P = pa.get_default_particle(np.float64).add_variable([pa.Variable("age", dtype=np.float32, initial=0), pa.Variable("temp", dtype=np.float32, initial=0)])
    pset = pa.ParticleSet(fs, pclass=P, x=case["x"], y=case["y"], z=case["z"], sort_by_cell=True)
    dt = float(case["dt"])
    pset.execute(kernels, runtime=2 * dt, dt=dt)  # warm-up: compile / load, cell sort
    t0 = time.perf_counter()
    pset.execute(kernels, runtime=steps * dt, dt=dt)
    wall = time.perf_counter() - t0
    st = pset._last_stats or {}
    return {"wall_s": wall, "steps_per_s_wall": n * steps / wall, "kernel_ms": st.get("kernel_ms"), "launches": st.get("launches"),
            "hosted": bool(st.get("hosted")), "program": st.get("program"), "remaining": len(pset), "jit_report": pset._kernel.jit_report,
            "temp_sum": float(np.sum(pset._data["temp"], dtype=np.float64))}


if __name__ == "__main__":
    import parcels_amd as pa

    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    steps = 24
    out = {"particles": n, "steps": steps,
           "rk4_alone": run([pa.AdvectionRK4], True, n, steps),
           "rk4_age_delete_compiled": run([pa.AdvectionRK4, Age, DeleteOld], True, n, steps),
           "rk4_age_delete_host_path": run([pa.AdvectionRK4, Age, DeleteOld], False, n, steps),
           "rk4_sample_compiled": run([pa.AdvectionRK4, SampleT], True, n, steps),
           "rk4_sample_host_path": run([pa.AdvectionRK4, SampleT], False, max(n // 5, 1), steps)}
    if os.environ.get("PARCELS_AMD_JIT_FAST_WAVES_AB"):
        for w in os.environ["PARCELS_AMD_JIT_FAST_WAVES_AB"].split(","):
            os.environ["PARCELS_AMD_JIT_FAST_WAVES"] = w
            out["rk4_sample_compiled_waves" + w] = run([pa.AdvectionRK4, SampleT], True, n, steps)
    print(json.dumps(out))
