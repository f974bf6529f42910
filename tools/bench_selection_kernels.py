#!/usr/bin/env python
"""What user kernels over SELECTIONS of the particles cost (parcels_amd/jit.py: conditional samples, masked stores): the kernels of
tests/test_gpu_jit_kernels.py -- the Argo-float state machine of tutorial_Argofloats.ipynb, a second sample only for the particles a first
one selects (tutorial_unstuck_Agrid.ipynb), the hand-written mid-point advection scheme of the tutorials -- on that file's small A-grid
FieldSet (24 x 18 nodes, 3 levels: everything cache-resident, so this prices the KERNELS, not the memory system), N particles, 24 steps:
compiled into the fused launch vs the host path (the reference's loop on the host columns, N / 10 particles), next to the built-in kernels.
Prints one JSON object.   python tools/bench_selection_kernels.py [N]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(kernels, jit, n, steps=24, dt=600.0):
    import parcels_amd as pa
    import test_gpu_jit_kernels as G

    os.environ["PARCELS_AMD_JIT"] = "1" if jit else "0"
    fs = G._fieldset("flat")
    rng = np.random.default_rng(1)
    lon, lat = np.asarray(fs.U.grid.lon), np.asarray(fs.U.grid.lat)
    x = lon[0] + (0.15 + 0.7 * rng.uniform(size=n)) * (lon[-1] - lon[0])
    y = lat[0] + (0.15 + 0.7 * rng.uniform(size=n)) * (lat[-1] - lat[0])
    pset = pa.ParticleSet(fs, pclass=G._pclass(np.float64), x=x, y=y, t=np.zeros(n))
    pset.execute(kernels, runtime=2 * dt, dt=dt)  # warm-up: compile / load
    t0 = time.perf_counter()
    pset.execute(kernels, runtime=steps * dt, dt=dt)
    wall = time.perf_counter() - t0
    st = pset._last_stats or {}
    return {"particles": n, "wall_s": wall, "particle_steps_per_s_wall": n * steps / wall, "kernel_ms": st.get("kernel_ms"), "launches": st.get("launches"),
            "hosted": bool(st.get("hosted")), "program": st.get("program"), "jit_report": (pset._kernel.jit_report or "")[:120]}


if __name__ == "__main__":
    import parcels_amd as pa
    import test_gpu_jit_kernels as G

    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
    out = {"fieldset": "tests/test_gpu_jit_kernels._fieldset('flat'): 24 x 18 A-grid, 3 levels, fp64", "steps": 24,
           "rk4_alone": run([pa.AdvectionRK4], True, n),
           "rk2_builtin": run([pa.AdvectionRK2], True, n),
           "rk2_user_written_compiled": run([G.MidpointAdvection, G.Age], True, n),
           "rk2_user_written_host_path": run([G.MidpointAdvection, G.Age], False, n // 10),
           "rk4_warm_water_drift_compiled": run([pa.AdvectionRK4, G.WarmWaterDrift], True, n),
           "rk4_warm_water_drift_host_path": run([pa.AdvectionRK4, G.WarmWaterDrift], False, n // 10),
           "argo_cycle_rk4_compiled": run([G.ArgoCycle, pa.AdvectionRK4], True, n),
           "argo_cycle_rk4_host_path": run([G.ArgoCycle, pa.AdvectionRK4], False, n // 10)}
    print(json.dumps(out))
