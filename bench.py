#!/usr/bin/env python
"""bench.py -- RK4 particle-steps/s of the HIP advection path on N MI355X GPUs (BASELINE.json metric).

Workload (BASELINE.json configs[1], "C2"): 3-D rectilinear A-grid 360 x 180 x 50 x 24 (lon, lat, depth, daily
levels), fp64 U and V, spherical mesh, 1e7 fp64 particles per GPU, AdvectionRK4, dt = 1 h.  One *step* = one RK4 dt
of every particle (4 velocity evaluations each: time search, 3 x 1-D cell search, 2 x 16-corner gather, 4-D linear
interpolation).  Fields and particles are resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 24 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Multi-GPU (weak scaling): particles are sharded by id, fields replicated, NO collective on the data path, so the timed K
steps contain none.  The only exchange of the path -- the periodic trajectory write-out, one RCCL all-gather of the output
columns (t, z, y, x, particle_id) over xGMI -- is executed once after the timed steps and reported separately
(`writeout_allgather_ms`, and `value_incl_writeout` = throughput if a write-out followed every K steps).

Prints ONE JSON line (rank 0).  `roofline`: achieved = algorithmic bytes per particle-step (1112 B for
C2 RK4 = 4 stages x 2 fields x 16 corners x 8 B + 88 B state) x particle-steps / advection-kernel time measured with
HIP events on the compute stream.  `cpu_baseline`: the scalar C oracle (oracle/parcels_oracle.c, "port") with OpenMP
on the host cores, on a bounded sample of the same workload.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP_C2_RK4 = 4 * 2 * 16 * 8 + 88
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md


def c2_case(npart: int, seed: int, nx=360, ny=180, nz=50, nt=24):
    """Synthetic C2 FieldSet: smooth analytic (Rossby-wave-like) U, V in m/s, |u| <= ~1 m/s, fp64."""
    lon = np.linspace(0.0, 360.0, nx)
    lat = np.linspace(-80.0, 80.0, ny)
    depth = np.linspace(0.0, 5000.0, nz)
    time_s = np.arange(nt) * 86400.0
    lam = np.deg2rad(lon)[None, None, None, :]
    phi = np.deg2rad(lat)[None, None, :, None]
    zz = (depth / 5000.0)[None, :, None, None]
    tt = (time_s / (nt * 86400.0))[:, None, None, None]
    U = (0.6 * np.cos(phi) * (1 - 0.5 * zz) + 0.3 * np.sin(3 * lam + 2 * np.pi * tt) * np.cos(2 * phi) * np.exp(-2 * zz)
         + 0.1 * np.cos(5 * lam - 4 * np.pi * tt) * np.sin(4 * phi))
    V = (0.3 * np.cos(3 * lam + 2 * np.pi * tt) * np.sin(2 * phi) * np.exp(-2 * zz) + 0.1 * np.sin(5 * lam - 4 * np.pi * tt) * np.cos(phi))
    rng = np.random.default_rng(seed)
    return dict(
        name="C2", mesh="spherical", lon=lon, lat=lat, depth=depth, x_pad="low", y_pad="low", z_pad="both", time_s=time_s,
        fields={"U": np.ascontiguousarray(U), "V": np.ascontiguousarray(V)},
        field_dims={"U": ("time", "depth", "YG", "XG"), "V": ("time", "depth", "YG", "XG")}, cgrid=False,
        kernels=["AdvectionRK4"], spatial_dtype="float64",
        x=rng.uniform(5.0, 355.0, npart), y=rng.uniform(-75.0, 75.0, npart), z=rng.uniform(10.0, 4990.0, npart),
        t0=None, dt=3600.0, runtime=None, seed=seed,
    )


def cpu_baseline(case, steps: int, sample: int):
    """Scalar C port (oracle) with OpenMP over particles on the host cores, bounded sample of the same workload."""
    from oracle import c_oracle as co

    c = dict(case)
    c["x"], c["y"], c["z"] = case["x"][:sample], case["y"][:sample], case["z"][:sample]
    c["runtime"] = steps * case["dt"]
    cores = os.cpu_count() or 1
    mc = co.MarshalledCase(c)
    data = co.initial_particles(c, mc.ngrids)
    data["dt"][:] = c["dt"]
    t0 = time.perf_counter()
    st = co.execute(mc, data, kernels=c["kernels"], endtime=c["runtime"], dt0=c["dt"], nthreads=cores)
    el = time.perf_counter() - t0
    return {"value": st["steps"] / el, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "sample": f"{sample} particles x {steps} RK4 steps of the same FieldSet, oracle/parcels_oracle.c with OpenMP ({el:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles", type=float, default=1e7, help="particles per GPU")
    ap.add_argument("--sort", type=int, default=1, help="cell-sort the device copy of the particles")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="particles of the CPU-baseline sample (~10-20 s on the host cores)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # rehearsal knob for a 1-GPU box: all ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device); the
    # line it prints is marked and is not a measurement
    rehearsal = os.environ.get("PARCELS_AMD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import parcels_amd as pa
    from tests.case_utils import build_fieldset, build_pset

    npart = int(args.particles)
    K, W = args.steps, args.warmup
    case = c2_case(npart, seed=1 + rank)
    nt = len(case["time_s"])
    if (K + W) * case["dt"] > case["time_s"][-1]:
        raise SystemExit(f"steps+warmup must stay within the {nt}-level time interval")
    fs = build_fieldset(case)
    fs.to_device(device=local_rank)
    pset = build_pset(case, fs, sort_by_cell=bool(args.sort))
    pset._data["particle_id"] += rank * npart  # shard by id
    kern = pa.Kernel([pa.AdvectionRK4], pset)
    eng = fs._engine_or_create()
    dt = case["dt"]
    pset._data["dt"][:] = dt
    eng.bind_particles(pset._data)
    torch.cuda.synchronize()
    t_h2d = time.perf_counter()
    eng.h2d()
    t_h2d = time.perf_counter() - t_h2d

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # measured device-to-device copy bandwidth: the practical HBM ceiling used as a second roofline denominator
    copy_gbps = eng.ctx.copy_bandwidth(1 << 30, 5) if rank == 0 else 0.0

    # warmup: W steps (also pays the one-off cell sort)
    if W > 0:
        eng.execute(kern.kernel_ids, endtime=W * dt, dt0=dt, sort_by_cell=int(args.sort), t_start=0.0)
    sync()
    t0 = time.perf_counter()
    st = eng.execute(kern.kernel_ids, endtime=(W + K) * dt, dt0=dt, sort_by_cell=0, t_start=W * dt)
    sync()
    el = time.perf_counter() - t0
    # the write-out exchange (not a step): all-gather of the output columns straight from the device columns
    t_ag = 0.0
    if dist is not None:
        from parcels_amd.distributed import allgather_output

        t1 = time.perf_counter()
        gathered = allgather_output(eng, world)
        sync()
        t_ag = time.perf_counter() - t1
        assert gathered["particle_id"].shape[0] == world * npart
        del gathered
    elt = torch.tensor([el], device="cuda", dtype=torch.float64)
    steps_t = torch.tensor([float(st["steps"])], device="cuda", dtype=torch.float64)
    kms = torch.tensor([st["kernel_ms"]], device="cuda", dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(elt, op=dist.ReduceOp.MAX)
        dist.all_reduce(steps_t, op=dist.ReduceOp.SUM)
        dist.all_reduce(kms, op=dist.ReduceOp.MAX)
    el = float(elt.item())
    total_steps = float(steps_t.item())
    agt = torch.tensor([t_ag], device="cuda", dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(agt, op=dist.ReduceOp.MAX)
    t_ag = float(agt.item())
    t_d2h = time.perf_counter()
    eng.d2h()
    t_d2h = time.perf_counter() - t_d2h
    ok = bool(np.all(pset._data["state"] == pa.StatusCode.EndofLoop))

    if rank == 0:
        value = total_steps / el
        kernel_s = float(kms.item()) * 1e-3
        per_gpu_steps = total_steps / world
        achieved = ALGO_BYTES_PER_STEP_C2_RK4 * per_gpu_steps / kernel_s / 1e9 if kernel_s > 0 else 0.0
        out = {
            "metric": "RK4 particle-steps/sec",
            "value": value,
            "unit": "particle-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": el / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C2: 3D rectilinear A-grid 360x180x50x24 fp64 U,V (spherical), AdvectionRK4, dt=3600 s",
                       "particles_per_gpu": npart, "particle_dtype": "f64", "cell_sorted": bool(args.sort),
                       "parallelism": f"particles sharded by id x{world}, fields replicated",
                       "all_states_endofloop": ok, **({"rehearsal_shared_gpu_gloo": True} if rehearsal else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": None, "kernel": "advect_kernel<double,0,0,RK4,lds>", "kernel_ms_per_launch": float(kms.item()),
                         "algorithmic_bytes_per_particle_step": ALGO_BYTES_PER_STEP_C2_RK4,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_STEP_C2_RK4 * per_gpu_steps,
                         "measured_copy_gbps": copy_gbps, "frac_of_measured_copy": achieved / copy_gbps if copy_gbps else None},
            # boundary costs outside the timed steps (rank 0): host<->device copies of the particle columns, write-out exchange
            "host_boundary": {"h2d_ms": t_h2d * 1e3, "d2h_ms": t_d2h * 1e3,
                              "value_pcie_inclusive": total_steps / (el + t_h2d + t_d2h)},
            "writeout_allgather_ms": t_ag * 1e3 if world > 1 else None,
            "value_incl_writeout": total_steps / (el + t_ag) if world > 1 else None,
        }
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):  # HBM bytes per launch from the rocprofv3 PMC passes (tools/pmc_summary.py), same command line
            try:
                pj = json.load(open(pmc))
                if pj.get("particles_per_gpu") == npart and pj.get("steps") == K:
                    out["roofline"]["traffic"] = pj["traffic_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = pj.get("source")
            except Exception:
                pass
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(case, steps=K, sample=min(args.cpu_sample, npart))
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
