"""Debug helper (GPU box): the staggered-release case of tests/test_gpu_fast_cgrid.py through the fast C-grid kernel, the general
program and the oracle; prints the particles on which they differ."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from case_utils import run_oracle  # noqa: E402
from test_gpu_fast_cgrid import _run  # noqa: E402

from oracle import cases  # noqa: E402

case = cases.curv_cgrid_case("fastc_ring", mesh="spherical", kernels=["AdvectionRK4_3D", "DeleteParticle"], seed=13, nt=6, npart=4000, dt=3600.0,
                             runtime=None, vel=1.5)
n = len(case["x"])
case["t0"] = np.random.default_rng(2).uniform(0, 3 * 86400.0, n)
case["t0"][::5] = 86400.0 * (np.arange(len(case["t0"][::5])) % 3)
case["endtime"] = 4.5 * 86400.0
case["runtime"] = None
mode = sys.argv[1] if len(sys.argv) > 1 else "stagger"
if mode == "same":
    case["t0"] = None
fast, ferr, fst = _run(case, True, endtime=case["endtime"])
gen, gerr, gst = _run(case, False, endtime=case["endtime"])
ref, oerr, _ = run_oracle(dict(case, populate=True), endtime=case["endtime"])
print("errors", ferr, gerr, oerr, "steps", fst["steps"], gst["steps"], "n", len(fast["x"]), len(gen["x"]), len(ref["x"]), "program", fst["program"], gst["program"])
ids = {k: set(d["particle_id"].tolist()) for k, d in (("fast", fast), ("gen", gen), ("ref", ref))}
print("only in fast vs ref", sorted(ids["fast"] - ids["ref"])[:20], "missing in fast", sorted(ids["ref"] - ids["fast"])[:20])
print("only in gen vs ref", sorted(ids["gen"] - ids["ref"])[:20], "missing in gen", sorted(ids["ref"] - ids["gen"])[:20])
t0 = case["t0"] if case["t0"] is not None else np.zeros(n)
for nm, d in (("fast", fast), ("gen", gen)):
    common = np.isin(d["particle_id"], ref["particle_id"])
    rc = np.isin(ref["particle_id"], d["particle_id"])
    dd = {k: d[k][common] for k in ("x", "y", "z", "t", "state", "ei", "particle_id")}
    rr = {k: ref[k][rc] for k in ("x", "y", "z", "t", "state", "ei", "particle_id")}
    bad = np.where((np.abs(dd["x"] - rr["x"]) > 1e-9) | (dd["t"] != rr["t"]) | (dd["state"] != rr["state"]))[0]
    print(nm, "differs from ref on", len(bad), "of", len(dd["x"]))
    for b in bad[:8]:
        pid = int(dd["particle_id"][b])
        print("  id", pid, "t0", t0[pid], "x", dd["x"][b], rr["x"][b], "t", dd["t"][b], rr["t"][b], "state", dd["state"][b], rr["state"][b], "ei", dd["ei"][b], rr["ei"][b])
for missing in sorted(ids["ref"] - ids["gen"])[:8]:
    print("  gen lost id", missing, "t0", t0[missing], "x0", case["x"][missing], case["y"][missing], case["z"][missing])
