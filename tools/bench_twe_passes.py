#!/usr/bin/env python
"""What a run that overshoots the last time level costs: the seed-9501 shape (AdvectionRK45 + DeleteOutOfBounds on a
2-level A-grid, runtime 4 h past the last level: ~100 samples fail call-wide) tiled to N particles -- passes over the call, wall seconds,
kernel milliseconds, with all failing samples listed per pass (round 5) and with one per pass (round 4).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_fuzz as fz
    from case_utils import run_hip

    from parcels_amd.engine import DeviceEngine

    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
    case, _ = fz.draw_case(9501)
    tiles = max(n // len(case["x"]), 1)
    for k in ("x", "y", "z"):
        case[k] = np.tile(np.asarray(case[k]), tiles)
    out = {"workload": f"fuzz seed 9501 tiled x{tiles} = {len(case['x'])} particles, {case['kernels']}, runtime {case['runtime']} s over a {case['time_s'][-1]:.0f} s interval"}
    for label, spec in (("all_keys_per_pass", 64), ("one_key_per_pass", 0)):
        if label == "one_key_per_pass" and len(sys.argv) > 2 and sys.argv[2] == "fast":
            continue
        DeviceEngine.TWE_SPECULATIVE_PASSES = spec
        run_hip(dict(case, x=case["x"][:662], y=case["y"][:662], z=case["z"][:662]))  # warm-up (module load, hash of nothing)
        t0 = time.perf_counter()
        got, err, st = run_hip(case)
        wall = time.perf_counter() - t0
        out[label] = {"passes": st["reran"] + 1, "failing_samples": len(st["time_error_keys"]), "wall_s": wall, "kernel_ms_last_pass": st["kernel_ms"],
                      "error": err, "survivors": int(len(got["x"]))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
