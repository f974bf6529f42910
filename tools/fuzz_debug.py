#!/usr/bin/env python
"""Re-run one fuzz seed (tests/test_gpu_fuzz.py) and print where HIP and the oracle differ."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as fz  # noqa: E402
from case_utils import run_hip, run_oracle  # noqa: E402

seed = int(sys.argv[1])
case, sort = fz.draw_case(seed)
ref, oerr, _ = run_oracle(case)
got, gerr, st = run_hip(case, sort_by_cell=sort)
print("seed", seed, case["kernels"], "errors", oerr, gerr, "n", len(ref["x"]), len(got["x"]), "env", {k: v for k, v in os.environ.items() if k.startswith("PK_")})
if len(ref["x"]) == len(got["x"]):
    for k in ("x", "y", "z", "t", "state", "ei"):
        a, b = np.asarray(got[k], dtype=np.float64), np.asarray(ref[k], dtype=np.float64)
        bad = np.flatnonzero(~np.isclose(a, b, rtol=float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6, atol=0, equal_nan=True).reshape(len(a), -1).all(axis=1))
        print(k, "differs at", bad[:10], [(got[k][i], ref[k][i]) for i in bad[:4]])
    i = int(sys.argv[3]) if len(sys.argv) > 3 else None
    if i is not None:
        print("particle", i, "start", case["x"][i], case["y"][i], case["z"][i], "got", got["x"][i], got["y"][i], got["z"][i], got["ei"][i], "ref", ref["x"][i], ref["y"][i], ref["z"][i], ref["ei"][i])
