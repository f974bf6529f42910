"""Fixtures from the reference's v3-JIT regression data (TEST INFRASTRUCTURE; runs in the build container only).

tests/test_interpolation.py:297-378 of the reference advects 455 particles for 4 s with AdvectionRK4_3D + DeleteParticle
through random fields (tests/test_data/test_interpolation_data_random_<interp>.nc) and demands that lon/lat/z at the
observations t = 0, 1, 2, 3 s equal the trajectories Parcels v3's JIT (C) kernels wrote
(tests/test_data/test_interpolation_jit_<interp>.zarr) to atol 1e-6.  Those are the only golden *trajectories* the
reference holds for this path, so they pin both the oracle and the HIP kernels:

    python -m oracle.make_v3_golden        # -> tests/golden/v3jit_{linear,cgrid_velocity,freeslip}.npz

Each fixture holds the inputs decoded from the .nc file (oracle/mini_hdf5.py), the v3 observations decoded from the zarr
store (oracle/mini_zarr.py), and the particle set the v4 reference (run here under oracle/ref_shim.py) ends with at t = 3 s.
The "nearest" variant uses an interpolator class defined inside the reference's test file, not in the library: skipped.
"""

from __future__ import annotations

import os
import sys

import numpy as np

sys.dont_write_bytecode = True

from . import make_golden as mg  # noqa: E402
from . import mini_hdf5, mini_zarr  # noqa: E402

TEST_DATA = "/root/reference/tests/test_data"
INTERPS = {"linear": dict(cgrid=False, slip=None), "cgrid_velocity": dict(cgrid=True, slip=None), "freeslip": dict(cgrid=False, slip="free")}


def v3_case(interp: str) -> dict:
    d = mini_hdf5.read(os.path.join(TEST_DATA, f"test_interpolation_data_random_{interp}.nc"))
    v3 = mini_zarr.read_group(os.path.join(TEST_DATA, f"test_interpolation_jit_{interp}.zarr"))
    order = np.argsort(v3["trajectory"], kind="stable")  # "v3 zarr is not sorted by particle_id" (test_interpolation.py:370)
    # release positions of the reference test (test_interpolation.py:349)
    x, y, z = np.meshgrid(np.linspace(0, 1, 7), np.linspace(0, 1, 13), np.linspace(0, 1, 5))
    dims = ("time", "depth", "YG", "XG")
    opts = INTERPS[interp]
    case = dict(
        name=f"v3jit_{interp}", mesh="flat",
        # "Convert the coordinates to float32 to match v3 behavior" (test_interpolation.py:313-315)
        lon=d["lon"].astype(np.float32), lat=d["lat"].astype(np.float32), depth=d["depth"].astype(np.float32),
        x_pad="low", y_pad="low", z_pad="high", time_s=d["time"].astype(np.float64),
        fields={"U": d["U"], "V": d["V"], "W": d["W"]}, field_dims={"U": dims, "V": dims, "W": dims},
        cgrid=opts["cgrid"], kernels=["AdvectionRK4_3D", "DeleteParticle"], spatial_dtype="float32",
        x=x.ravel().astype(np.float64), y=y.ravel().astype(np.float64), z=z.ravel().astype(np.float64),
        t0=None, dt=1.0, runtime=3.0, seed=0,
        v3_lon=np.asarray(v3["lon"][order], dtype=np.float64), v3_lat=np.asarray(v3["lat"][order], dtype=np.float64),
        v3_z=np.asarray(v3["z"][order], dtype=np.float64), v3_time=np.asarray(v3["time"][order], dtype=np.float64),
    )
    if opts["slip"]:
        case["slip"] = opts["slip"]
    return case


def observations(run, case):
    """lon/lat/z (n, 4) at t = 0..3 s from `run(case, endtime) -> soa dict`; deleted particles are NaN, as in the v3 file."""
    n = len(case["x"])
    obs = {k: np.full((n, 4), np.nan) for k in ("x", "y", "z")}
    sdt = np.dtype(case["spatial_dtype"])
    for k, src in (("x", "x"), ("y", "y"), ("z", "z")):
        obs[k][:, 0] = np.asarray(case[src]).astype(sdt)
    for step in (1, 2, 3):
        out = run(case, float(step))
        ids = np.asarray(out["particle_id"])
        for k in ("x", "y", "z"):
            obs[k][ids, step] = out[k]
    return obs


def main():
    for interp in INTERPS:
        case = v3_case(interp)

        def run(c, endtime):
            cc = dict(c)
            cc["runtime"] = endtime
            return mg.ref_run_case(cc)[0]

        obs = observations(run, case)
        worst = 0.0
        for k, ref in (("x", "v3_lon"), ("y", "v3_lat"), ("z", "v3_z")):
            assert np.array_equal(np.isnan(obs[k]), np.isnan(case[ref])), f"{interp}: deleted sets differ on {k}"
            worst = max(worst, float(np.nanmax(np.abs(obs[k] - case[ref]))))
        assert worst <= 1e-6, worst  # the reference's own bar (test_interpolation.py:376-378)
        out, err, extras = mg.ref_run_case(case)
        path = os.path.join(mg.GOLDEN_DIR, case["name"] + ".npz")
        mg.save_case(path, case, out, err, extras)
        print(f"{case['name']:24s} n={len(out['x'])} of {len(case['x'])} alive at t=3, v4-under-shim vs v3 JIT: max |diff| = {worst:.2e}, "
              f"{os.path.getsize(path) // 1024} KB")


if __name__ == "__main__":
    main()
