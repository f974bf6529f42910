#!/usr/bin/env python
"""bench.py -- RK4 particle-steps/s of the HIP advection path on N MI355X GPUs (BASELINE.json metric).

Workload (BASELINE.json configs[1], "C2"): 3-D rectilinear A-grid 360 x 180 x 50 x 24 (lon, lat, depth, daily
levels), fp64 U and V, spherical mesh, 1e7 fp64 particles per GPU, AdvectionRK4, dt = 1 h.  One *step* = one RK4 dt
of every particle (4 velocity evaluations each: time search, 3 x 1-D cell search, 2 x 16-corner gather, 4-D linear
interpolation).  Fields and particles are resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 24 --warmup 2
    python bench.py --gpus N ...          (without a launcher: starts its own N ranks under torch.distributed.run, 127.0.0.1 rendezvous)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Multi-GPU (weak scaling, 1e7 particles per GPU): ONE id space of N x 1e7 particles, generated in shard-independent blocks and
sharded by id (parcels_amd.distributed.shard_slice), fields replicated, NO collective on the data path, so the timed K steps
contain none.  The only exchange of the path -- the periodic trajectory write-out, one RCCL all-gather of the to-write
columns (t, z, y, x, particle_id) over xGMI -- is executed once after the timed steps and reported separately
(`writeout_allgather_ms`, and `value_incl_writeout` = throughput if a write-out followed every K steps).

Prints ONE JSON line (rank 0).  What the objects mean:

`roofline` -- the advection kernel is bound by fp64-rate VALU issue, not by HBM (its gathers are served by L2 / Infinity Cache):
  bound     "valu_fp64"
  achieved  ALGORITHMIC fp64 TFLOP/s = 143.46 flops per velocity evaluation (ALGO_FLOPS_PER_EVAL_C2 below) x 4 evaluations x particle-steps
            of the timed launch / kernel time measured HERE with HIP events on the compute stream;  peak = 78.6 TFLOP/s (fp64 vector FMA);
            frac = achieved / peak.  `valu_busy_frac` (what rounds 1-4 printed as `frac`) = SQ_ACTIVE_INST_VALU x 4 per particle-step of
            the rocprofv3 PMC pass in profiles/pmc_latest.json x particle-steps / kernel time / (1024 SIMDs x 2.4 GHz): utilisation, not a
            roof.  `sclk_mhz` = the shader clock of ONE MORE, untimed repetition of the timed launch (a cycle-counter probe spins beside that one only)
  traffic   HBM bytes of the timed launch (FETCH_SIZE + WRITE_SIZE passes, calibrated on the 1 GiB copy kernel), scaled per
            particle-step to this run;  `hbm` = that traffic over the kernel time against the 8 TB/s peak
  algorithmic  SURVEY.md 8(d)'s byte model (1112 B per particle-step = 4 stages x 2 fields x 16 corners x 8 B + 88 B state) over
            the kernel time.  It exceeds the HBM peak because those bytes come out of cache: it is NOT an HBM fraction.
`secondary` (N = 1 only; `--secondary 0` switches it off) -- BASELINE configs 3 and 5 at FULL size next to the headline: C3 AdvectionRK4_3D,
  C5 AdvectionRK45 and AdvectionDiffusionM1 on the 4322 x 3059 x 75 curvilinear C-grid with 1e7 particles, each with kernel ms, value,
  a `roofline` by SURVEY 8(d)'s algorithmic bytes (536 / 672 / 568 B per unit; these working sets ARE beyond the caches) plus the counter
  traffic of profiles/pmc_secondary_latest.json, and `check`: 1e5 particle ids re-run through the CPU oracle on the same arrays.
`check` -- the first 1e5 particle ids after the timed steps, re-run alone through the CPU oracle (test infrastructure; `--check 0` switches it off).
`with_output` (N = 1 only; `--with-output 0` switches it off) -- the headline workload with a ParticleFile every 24 steps: value_incl_output,
  output_hidden_frac (tools/bench_writeout.py has the sweep over cadences).
`user_kernels` (N = 1 only; `--user-kernels 0` switches it off) -- AdvectionRK4 + two user-written Python kernels on the headline FieldSet:
  compiled into the launch (parcels_amd/jit.py) vs the host path, 2e6 particles, 24 steps, wall seconds.
`repeat_execute` (N = 1 only) -- the headline steps as ten consecutive pset.execute calls: the particle columns stay device-resident between them
  (parcels_amd/columns.py); wall per later call, kernel ms, what crossed PCIe per call.
`batch_agreements` (N > 1 only) -- four more steps with the agreements of a sharded ParticleSet installed (parcels_amd.distributed.batch_agreement):
  all-reduce calls per rank, seconds inside them, wall and kernel ms (max over ranks); outside `value`.
`cpu_baseline` -- oracle/fast_agrid_cpu.c (kind "port"): the headline workload restated the way one writes it for a CPU, bit-identical
  to the checker oracle, OpenMP on this box's host cores, bounded sample.
`cpu_baseline_reference` -- the reference itself (Parcels under oracle/ref_shim.py).  It is Python and /root/reference does not exist
  on the GPU box, so this leg is OFF-BOX: timed by tools/time_reference_cpu.py in the build container (profiles/r02_cpu_reference.json,
  an 8-core Xeon) and attached with that label.
"""

from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP_C2_RK4 = 4 * 2 * 16 * 8 + 88
# fp64 flops one velocity evaluation of the reference's algorithm asks for (an FMA counts 2): time / depth / lat / lon barycentric
# coordinates, the t- and z-lerps of 2 x 16 corners, the bilinear sums, the metres -> degrees conversion with one cosine -- counted on
# the kernel that executes every one of them (PMC instruction count x static instruction mix of the round-4 binary,
# profiles/r04q_c2_pmc.md x profiles/r03_c2_isa.json).  ALGORITHMIC like the byte model: the corner-block cache of
# round 5 executes fewer (it re-uses the lerped corners between stages that share t), which must not lower the work it is credited with.
ALGO_FLOPS_PER_EVAL_C2 = 143.46
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 16 lanes x 2 (FMA) x 2.4 GHz
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md
N_SIMD, PEAK_CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs; one fp64-rate VALU instruction occupies a SIMD for 4 cycles
POS_BLOCK = 1 << 20  # particles per generator block (positions do not depend on how the id space is sharded)


def c2_positions(seed: int, lo: int, hi: int):
    """x, y, z of particles lo..hi-1 of the global id space: block b of POS_BLOCK ids is drawn from default_rng([seed, b])."""
    xs, ys, zs = [], [], []
    for b in range(lo // POS_BLOCK, (max(hi, lo + 1) - 1) // POS_BLOCK + 1):
        rng = np.random.default_rng([seed, b])
        x = rng.uniform(5.0, 355.0, POS_BLOCK)
        y = rng.uniform(-75.0, 75.0, POS_BLOCK)
        z = rng.uniform(10.0, 4990.0, POS_BLOCK)
        s = slice(max(lo - b * POS_BLOCK, 0), min(hi - b * POS_BLOCK, POS_BLOCK))
        xs.append(x[s]); ys.append(y[s]); zs.append(z[s])
    return np.concatenate(xs), np.concatenate(ys), np.concatenate(zs)


def c2_case(seed: int = 1, lo: int = 0, hi: int = 10_000_000, nx=360, ny=180, nz=50, nt=24):
    """Synthetic C2 FieldSet: smooth analytic (Rossby-wave-like) U, V in m/s, |u| <= ~1 m/s, fp64; particles lo..hi-1."""
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
    x, y, z = c2_positions(seed, lo, hi)
    return dict(
        name="C2", mesh="spherical", lon=lon, lat=lat, depth=depth, x_pad="low", y_pad="low", z_pad="both", time_s=time_s,
        fields={"U": np.ascontiguousarray(U), "V": np.ascontiguousarray(V)},
        field_dims={"U": ("time", "depth", "YG", "XG"), "V": ("time", "depth", "YG", "XG")}, cgrid=False,
        kernels=["AdvectionRK4"], spatial_dtype="float64", x=x, y=y, z=z, t0=None, dt=3600.0, runtime=None, seed=seed,
    )


def cpu_baseline(case, steps: int, sample: int):
    """The fair native CPU leg: oracle/fast_agrid_cpu.c -- the headline workload written the way one writes it for a CPU (hinted
    searches, no dtype emulation, OpenMP over cell-sorted particles), held bit-identical to the checker oracle by
    tests/test_oracle_fast_cpu.py -- on this box's host cores, bounded sample of the same workload.  The cell sort of the sample is
    outside the timed call, like the GPU's.  (The reference itself is Python and cannot travel to the GPU box: see
    `cpu_baseline_reference`.)"""
    from oracle import c_oracle as co

    c = dict(case)
    c["x"], c["y"], c["z"] = case["x"][:sample], case["y"][:sample], case["z"][:sample]
    cores = os.cpu_count() or 1
    best = None
    for threads in sorted({cores, max(cores // 2, 1)}):  # SMT siblings do not always help a gather-bound loop: report the better of the two
        _, nsteps, el = co.fast_rk4_agrid(c, endtime=steps * case["dt"], nthreads=threads, sort_by_cell=True)
        if best is None or nsteps / el > best[0]:
            best = (nsteps / el, threads, el)
    return {"value": best[0], "unit": "particle-steps/s", "cores": best[1], "kind": "port",
            "sample": f"{sample} particles x {steps} RK4 steps of the same FieldSet, oracle/fast_agrid_cpu.c (OpenMP, cell-sorted, bit-identical to "
                      f"the checker oracle) on {best[1]} of {cores} hardware threads ({best[2]:.1f} s)"}


def check_headline(case, data, n_check: int, endtime: float):
    """The first `n_check` particle ids (of this rank's shard) after the timed steps against oracle/parcels_oracle.c run on those particles
    alone -- the checker, not the thing measured."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from case_utils import compare, run_oracle

    sub = dict(case)
    sub["x"], sub["y"], sub["z"] = case["x"][:n_check], case["y"][:n_check], case["z"][:n_check]
    sub["runtime"] = endtime
    t0 = time.perf_counter()
    ref, err, _ = run_oracle(sub, nthreads=os.cpu_count() or 1)
    oracle_s = time.perf_counter() - t0
    assert err is None, f"oracle raised {err}"
    ids = np.asarray(data["particle_id"])
    lo = int(ids.min()) if len(ids) else 0
    sel = np.flatnonzero((ids >= lo) & (ids < lo + n_check))
    sel = sel[np.argsort(ids[sel], kind="stable")]  # host rows are in id order already; a shard's ids start at its offset
    got = {k: np.asarray(v)[sel] for k, v in data.items()}
    got["particle_id"] = got["particle_id"] - lo
    rep = compare(got, ref, rtol=1e-12, check_state="all", label="headline subset vs oracle", skip=())
    return {"passed": True, "n_check": int(n_check), "max_rel_diff": rep, "tolerance": "1e-12 relative on x, y, z; state, ei, t, ids exact",
            "steps": int(round(endtime / case["dt"])), "oracle_s": oracle_s}


def self_launch(n: int, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks under torch.distributed.run (one process per GPU, rendezvous
    on 127.0.0.1 at a free port) and pass their output through -- rank 0 prints the one JSON line.  On a box with fewer than N
    GPUs the ranks share cuda:0 and talk over gloo (rehearsal; the line is marked `rehearsal_shared_gpu_gloo` and is no measurement)."""
    import socket
    import subprocess

    import torch

    env = dict(os.environ)
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        env["PARCELS_AMD_BENCH_REHEARSAL"] = "1"
        print(f"[bench] {have} GPU(s) visible, {n} ranks requested: rehearsal on cuda:0 over gloo", file=sys.stderr)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


def launch_command_world(argv_gpus: int, environ) -> tuple[int, bool]:
    """(world size the process will run with, must it launch the ranks itself) -- the rule main() applies, for the CPU test."""
    if argv_gpus > 1 and "WORLD_SIZE" not in environ:
        return argv_gpus, True
    return int(environ.get("WORLD_SIZE", "1")), False


ALGO_BYTES_PER_STEP_C3_RK4_3D = 4 * (12 * 4 + 8 * 8) + 88  # 12 staggered f32 values + 8 corner coordinates per evaluation
ALGO_BYTES_PER_STEP_C5_RK45 = 6 * (8 * 4 + 8 * 8) + 96      # per attempt: 6 evaluations of U, V at two levels + the corner coordinates
ALGO_BYTES_PER_STEP_C5_M1 = 568


def user_kernel_runs(n):
    """The reference's plug-in point #1 on the headline FieldSet: AdvectionRK4 followed by two user-written Python kernels (ageing,
    delete-when-old), (a) translated and compiled into the fused launch (parcels_amd/jit.py), (b) on the host path (the reference's loop on
    the host columns), next to (c) AdvectionRK4 alone; wall times include the H2D / D2H of the particle columns."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_user_kernels as bu
    import parcels_amd as pa

    steps = 24
    old = os.environ.get("PARCELS_AMD_JIT")
    try:
        res = {"particles": n, "steps": steps, "kernels": "AdvectionRK4, Age, DeleteOld",
               "rk4_alone": bu.run([pa.AdvectionRK4], True, n, steps),
               "compiled": bu.run([pa.AdvectionRK4, bu.Age, bu.DeleteOld], True, n, steps),
               "host_path": bu.run([pa.AdvectionRK4, bu.Age, bu.DeleteOld], False, n, steps)}
    finally:
        if old is None:
            os.environ.pop("PARCELS_AMD_JIT", None)
        else:
            os.environ["PARCELS_AMD_JIT"] = old
    res["same_survivors"] = res["compiled"]["remaining"] == res["host_path"]["remaining"]
    res["speedup_wall"] = res["host_path"]["wall_s"] / max(res["compiled"]["wall_s"], 1e-9)
    return res


def long_run(case, fs, steps):
    """The headline workload over ALL its time levels in one Kernel.execute (552 steps of 1 h through the 24 daily levels, one cell sort,
    one fused launch; DeleteParticle because a few particles reach the edge of the domain in 23 days): the headline's 9 ms window is 4 %
    of it.  Through the product path (ParticleSet.execute: H2D of the columns, sort, launch, D2H)."""
    import torch

    import parcels_amd as pa
    from tests.case_utils import build_pset

    pset = build_pset(case, fs, sort_by_cell=True, resort_every=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pset.execute([pa.AdvectionRK4, pa.DeleteParticle], dt=case["dt"], runtime=steps * case["dt"])
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    st = pset._last_stats
    ks = st["kernel_ms"] * 1e-3
    return {"workload": f"the headline FieldSet and particles, AdvectionRK4 + DeleteParticle, {steps} steps of {case['dt']:.0f} s through all "
                        f"{len(case['time_s'])} levels, one cell sort, {st['launches']} launch(es)",
            "particle_steps": int(st["steps"]), "kernel_ms": st["kernel_ms"], "cell_sort_ms": st["sort_ms"], "remaining_particles": len(pset),
            "value": st["steps"] / ks, "unit": "particle-steps/s (kernel time)", "value_incl_sort": st["steps"] / (ks + st["sort_ms"] * 1e-3),
            "value_end_to_end": st["steps"] / wall, "wall_s_incl_h2d_d2h": wall}


def repeat_execute(case, fs, calls, steps):
    """The usual script loop: `calls` x pset.execute(AdvectionRK4, runtime = steps x dt) on the headline FieldSet, through the product path.
    The particle columns stay device-resident between the calls (parcels_amd/columns.py): the first call uploads them and sorts, the others
    move nothing across PCIe unless the script touches the set -- what each call costs end to end, next to the kernel time of the same steps."""
    import torch

    import parcels_amd as pa
    from tests.case_utils import build_pset

    pset = build_pset(case, fs, sort_by_cell=True, resort_every=0)
    eng = fs._engine_or_create()
    walls, kms, moved = [], [], []
    nsteps = 0
    for k in range(calls):
        before = dict(eng.transfers)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pset.execute([pa.AdvectionRK4, pa.DeleteParticle], dt=case["dt"], runtime=steps * case["dt"])  # (DeleteParticle: like `long_run`, a few particles reach the edge)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        st = pset._last_stats
        kms.append(st["kernel_ms"])
        nsteps = int(st["steps"])
        moved.append({key: eng.transfers[key] - before[key] for key in before if eng.transfers[key] != before[key]})
    t0 = time.perf_counter()
    x_sum = float(np.sum(pset._data.peek("x") if hasattr(pset._data, "peek") else pset._data["x"]))  # the first host read: one column comes down
    t_read = time.perf_counter() - t0
    later = sorted(walls[1:])
    med = later[(len(later) - 1) // 2]
    kmed = sorted(kms[1:])[(len(kms[1:]) - 1) // 2]
    return {"workload": f"{calls} x ParticleSet.execute([AdvectionRK4, DeleteParticle], {steps} steps of {case['dt']:.0f} s) on the headline FieldSet and particles",
            "particle_steps_per_call": nsteps, "wall_ms_first_call": walls[0] * 1e3, "wall_ms_later_calls": {"min": later[0] * 1e3, "median": med * 1e3, "max": later[-1] * 1e3},
            "kernel_ms_later_calls_median": kmed, "value_end_to_end_later_calls": nsteps / med, "value_kernel_later_calls": nsteps / (kmed * 1e-3),
            "unit": "particle-steps/s", "pcie_transfers_per_call": moved, "first_host_read_of_x_ms": t_read * 1e3, "checksum_x": x_sum}


def with_output(case, fs, steps, every):
    """The headline workload WITH trajectory output: a ParticleFile every `every` steps over `steps` steps, three ways on
    fresh ParticleSets of the headline particles -- no output, the inline write-out (D2H of the to-write columns + filter + Parquet encode
    between two launches: what rounds 1-2 did and the reference does), the asynchronous one (device write filter -> pinned snapshot on the
    copy stream -> writer thread; parcels_amd/particlefile.py).  value_incl_output = particle-steps / wall of the asynchronous run;
    output_hidden_frac = the share of the inline write-out cost that no longer shows in the wall clock."""
    import tempfile

    import torch

    import parcels_amd as pa
    from tests.case_utils import build_pset

    dt = case["dt"]
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for mode in ("warmup", "none", "inline", "async"):
            pset = build_pset(case, fs, sort_by_cell=True, resort_every=0)
            pset.async_output = mode == "async"
            pf = None if mode == "none" else pa.ParticleFile(os.path.join(tmp, f"{mode}.parquet"), outputdt=float(every * dt))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pset.execute(pa.AdvectionRK4, dt=dt, runtime=(2 * every if mode == "warmup" else steps) * dt, output_file=pf)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            if mode == "warmup":
                continue  # (pins the snapshot buffers, starts the thread pools, first touch of the temp directory)
            res[mode] = {"wall_s": wall}
            if pf is not None:
                res[mode].update(file_MB=os.path.getsize(pf.path) / 1e6, parquet_writer_s=getattr(pf, "writer_seconds", None))
            if mode == "async":
                res[mode]["writer"] = getattr(pset, "_last_writer_stats", None)
            os.path.exists(os.path.join(tmp, f"{mode}.parquet")) and os.remove(os.path.join(tmp, f"{mode}.parquet"))
    n = len(case["x"])
    tables = steps // every + 1
    cost_inline = res["inline"]["wall_s"] - res["none"]["wall_s"]
    cost_async = res["async"]["wall_s"] - res["none"]["wall_s"]
    row_bytes = 8 * 5  # particle_id, t, z, y, x of the fp64 particle class
    return {"workload": f"headline FieldSet and particles, AdvectionRK4, {steps} steps, one ParticleFile table every {every} steps ({tables} tables of {n} rows, zstd)",
            "wall_s": {k: v["wall_s"] for k, v in res.items()}, "detail": res,
            "value_incl_output": n * steps / res["async"]["wall_s"], "value_no_output_same_run": n * steps / res["none"]["wall_s"], "unit": "particle-steps/s",
            "write_out_cost_s": {"inline": cost_inline, "async": cost_async}, "async_not_slower_than_inline": bool(res["async"]["wall_s"] <= res["inline"]["wall_s"] * 1.02),
            "output_hidden_frac": (1.0 - cost_async / cost_inline) if cost_inline > 0 else None,
            "table_GB_per_s_async": tables * n * row_bytes / 1e9 / max(cost_async, 1e-9),
            "note": "a table of 1e7 fp64 rows is 400 MB: 7 ms of PCIe alone against 8 ms of kernel per 24 steps -- at this cadence the run is bound by the "
                    "host's encode + file rate, not by the GPU; tools/bench_writeout.py sweeps the cadence"}


def secondary_runs(args):
    """BASELINE configs 3 and 5 at full size on this GPU (tools/bench_configs.py builds them), outside the timed region of the
    headline: per run the particle-steps/s of the fused launch (HIP events on the compute stream), the algorithmic-byte roofline
    fraction, the HBM counters of the last committed rocprofv3 PMC pass of that program, and the verdict of re-running the first
    `--secondary-check` particle ids through the CPU oracle on the same arrays (the oracle is the checker here, never the thing
    measured: tools/bench_configs.py::check_against_oracle)."""
    from tools import bench_configs as bc

    out = []
    algo = {"AdvectionRK4_3D": ALGO_BYTES_PER_STEP_C3_RK4_3D, "AdvectionRK45": ALGO_BYTES_PER_STEP_C5_RK45, "AdvectionDiffusionM1": ALGO_BYTES_PER_STEP_C5_M1}
    pmc = {}
    pj_path = os.path.join(ROOT, "profiles", "pmc_secondary_latest.json")
    if os.path.exists(pj_path):
        try:
            pmc = json.load(open(pj_path))
        except Exception:
            pmc = {}
    for config in ("c3", "c5"):
        t0 = time.perf_counter()
        try:
            res = bc.run_config(config, scale=args.secondary_scale, particles=args.secondary_particles, steps=24, nt=4, nslots=3, nz=75,
                                check=int(args.secondary_check), emit=lambda o: None, reps=int(args.secondary_reps))
        except AssertionError as e:  # the subset check failed: report it, do not hide it
            out.append({"config": config, "check": {"passed": False, "error": str(e)[:2000]}})
            continue
        for r in res:
            ks = r["kernel_ms"] * 1e-3  # (the MEDIAN of the timed launches: tools/bench_configs.py)
            units = r["attempts"] if r["kernels"] == "AdvectionRK45" else r["particle_steps"]  # RK45: bytes move per attempt
            if r["kernels"] == "AdvectionRK45":
                units = units / 2  # `attempts` counts the DeleteParticle slot of every attempt too
            ab = algo[r["kernels"]]
            e = {"config": config, "workload": f"{config.upper()}: curvilinear C-grid {r['grid'][0]}x{r['grid'][1]}x{r['grid'][2]} f32 U,V,W, "
                                               f"{r['nslots']}-slot ring, {r['particles']} fp64 particles, {r['kernels']} + DeleteParticle, 24 steps of 3600 s",
                 "kernels": r["kernels"], "particles": r["particles"], "particle_steps": r["particle_steps"], "attempts": r["attempts"], "kernel_ms": r["kernel_ms"],
                 "kernel_ms_stats": r.get("kernel_ms_stats"), "sclk_mhz": r.get("sclk_mhz"),
                 "value": r["particle_steps_per_s_kernel"], "unit": "particle-steps/s (kernel time, levels resident)",
                 "cell_sort_ms": r["sort_ms"], "wall_s_incl_h2d_d2h": r["wall_s"],
                 "roofline": {"bound": "hbm", "algorithmic_bytes_per_unit": ab, "units": units,
                              "achieved": ab * units / ks / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ab * units / ks / 1e9 / HBM_PEAK_GBPS,
                              "traffic": None},
                 "check": None, "setup_s": {"dataset": r["dataset_generation_s"], "device_create": r["device_create_s"]}}
            pp = pmc.get(r["kernels"])
            if pp and pp.get("hbm_bytes_per_particle_step") is not None:  # counters of the committed PMC pass, scaled to this launch
                tr = pp["hbm_bytes_per_particle_step"] * r["particle_steps"]
                e["roofline"]["traffic"] = tr
                e["roofline"]["hbm_counter_frac"] = tr / ks / 1e9 / HBM_PEAK_GBPS
                e["roofline"]["counters_source"] = pp.get("source")
                e["roofline"]["valu_insts_per_wave_evaluation"] = pp.get("valu_insts_per_wave_eval")
                e["roofline"]["scratch_bytes_per_lane"] = pp.get("scratch")
                e["roofline"]["counters_stale"] = counters_stale(r["kernels"], pp)  # the kernel changed since the PMC passes
            vp = r.get("velocity_pairs")
            if vp:  # the opt-in pair copies, timed: what a level pair costs with them (kernel + packing) against `kernel_ms` without
                e["velocity_pairs"] = dict(vp)
                if vp.get("kernel_plus_pack_ms"):
                    e["velocity_pairs"]["frac_kernel_only"] = ab * units / (vp["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
                    e["velocity_pairs"]["frac_incl_pack"] = ab * units / (vp["kernel_plus_pack_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
            if r.get("check"):
                c = r["check"]
                e["check"] = {"passed": True, "n_check": c["n_check"], "deleted": c["deleted"], "exact": c["exact"],
                              "max_abs_diff": c["max_abs_diff"], "tolerance": c["tolerance"], "oracle_s": c["oracle_s"],
                              "oracle_hash_table": c.get("oracle_hash_table")}
            out.append(e)
        out[-1]["total_s"] = time.perf_counter() - t0
    # The attainable roof of the kernel furthest from its roofline: tools/gather_roof -- the access pattern of the RK45
    # kernel WITHOUT its arithmetic at the same residency (cell changes with probability 0.141 per lane-evaluation, six dependent 128-byte
    # lines each, 88 + 96 B of state) -- run here on the same GPU, after the product's fields were released
    gr = os.path.join(ROOT, "tools", "gather_roof")
    if args.secondary_scale == 1.0:
        import gc
        import subprocess

        if not os.path.exists(gr):  # (normally built by __graft_entry__.build(); hipcc is on the GPU box too)
            subprocess.run(["bash", os.path.join(ROOT, "tools", "build_gather_roof.sh")], capture_output=True, timeout=300)
        gc.collect()
        for e in out:
            if e.get("kernels") != "AdvectionRK45" or not os.path.exists(gr):
                continue
            try:
                units = e["roofline"]["units"]
                per_particle = max(int(round(units / e["particles"])) if e.get("particles") else 5, 1)
                res = {}
                for label, alu in (("memory_only", 0), ("with_1000_dependent_fp64_fma_per_evaluation", 1000)):
                    txt = subprocess.run([gr, "--particles", str(int(args.secondary_particles)), "--attempts", str(per_particle), "--alu", str(alu)],
                                         capture_output=True, text=True, timeout=300).stdout.strip().split("\n")[-1]
                    res[label] = json.loads(txt)
                m = res["memory_only"]
                att_ms = m["ms_per_launch_median"] * units / m["attempts"]  # scaled to this launch's number of attempts
                e["roofline"]["attainable"] = {
                    "what": "tools/gather_roof.hip: the kernel's access pattern without its arithmetic, same GPU, same residency (12 one-wavefront workgroups per CU)",
                    "ms_for_this_launch": att_ms, "GBps_algorithmic": e["roofline"]["algorithmic_bytes_per_unit"] * units / (att_ms * 1e-3) / 1e9,
                    "frac_of_peak": e["roofline"]["algorithmic_bytes_per_unit"] * units / (att_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "kernel_ms_over_attainable_ms": e["kernel_ms"] / att_ms, "runs": res,
                    "reading": "the memory side alone would allow frac_of_peak; the kernel's distance from it is arithmetic (fp64 VALU issue + dependent "
                               "latency at 3 waves per SIMD), not bandwidth"}
            except Exception as ex:  # the bench line must survive a failing side measurement
                e["roofline"]["attainable"] = {"error": repr(ex)[:500]}
    return out


def kernel_source_hash():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "parcels_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "parcels_amd", "csrc", "*.hip"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def counters_stale(key, entry):
    """Were the PMC counters of `entry` (profiles/pmc_*latest.json) collected on another kernel than the one this library runs?  By the
    hash of the kernel's MACHINE CODE when both sides have one (tools/kernel_code_hash.py writes parcels_amd/kernel_code_hashes.json at
    build time, tools/pmc_summary.py records it with the counters): a source change that leaves the kernel's instructions alone does
    not make its counters stale.  Otherwise by the hash of all kernel sources."""
    try:
        cur = json.load(open(os.path.join(ROOT, "parcels_amd", "kernel_code_hashes.json"))).get(key, {}).get("code_hash")
    except Exception:
        cur = None
    if cur and entry.get("code_hash"):
        return cur != entry["code_hash"]
    return entry.get("source_hash") != kernel_source_hash()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reps", type=int, default=7,
                    help="the timed region (barrier, EXACTLY --steps steps, barrier) is repeated this many times, each from the device checkpoint taken "
                         "after the warm-up; `value`, `ms_per_step` and `roofline` use the MEDIAN repetition, min / max are reported next to it")
    ap.add_argument("--long-run", type=int, default=552,
                    help="N = 1 only: also run the headline workload over ALL its time levels (this many steps, one launch) as `long_run`; 0 = off")
    ap.add_argument("--repeat-execute", type=int, default=10,
                    help="N = 1 only: also run the headline steps as this many consecutive pset.execute calls (device-resident columns between the calls) as "
                         "`repeat_execute`; 0 = off")
    ap.add_argument("--secondary-reps", type=int, default=5, help="timed launches per secondary kernel list after one cold launch")
    ap.add_argument("--c4", type=float, default=0,
                    help="also run BASELINE config 4 after the headline, on the same ranks: the NEMO-size curvilinear C-grid with this many particles PER "
                         "GPU (1e7 -> 8e7 on 8 GPUs), AdvectionRK4_3D, one id space sharded by id, ParticleFile on rank 0 fed by the gather of the "
                         "to-write columns (tools/bench_configs.py::run_c4); attached as `c4`; 0 = off")
    ap.add_argument("--particles", type=float, default=1e7, help="particles per GPU")
    ap.add_argument("--sort", type=int, default=1, help="cell-sort the device copy of the particles")
    ap.add_argument("--with-output", type=int, default=96,
                    help="N = 1: the headline workload over this many steps with a ParticleFile every 24 steps -- no output / inline / asynchronous (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", type=float, default=1e5, help="headline: particle ids re-run through the CPU oracle after the timed steps (0 = off)")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="particles of the CPU-baseline sample (~10-20 s on the host cores)")
    ap.add_argument("--secondary", type=int, default=1,
                    help="N = 1 only: also run BASELINE configs 3 and 5 (C3 RK4_3D, C5 RK45 + M1) at full size, each with a subset "
                         "re-run through the CPU oracle, and attach them as `secondary` (0 = off)")
    ap.add_argument("--user-kernels", type=float, default=2e6,
                    help="N = 1 only, with --secondary: particles of the user-kernel leg (C2 FieldSet, AdvectionRK4 + a user-written ageing and a "
                         "delete-when-old kernel: compiled into the launch by parcels_amd/jit.py vs the host path); 0 = off")
    ap.add_argument("--secondary-check", type=float, default=1e5, help="particle ids of every secondary run re-run through the oracle")
    ap.add_argument("--secondary-scale", type=float, default=1.0, help="shrinks nx, ny of the secondary grid (1.0 = BASELINE size)")
    ap.add_argument("--secondary-particles", type=float, default=1e7)
    args = ap.parse_args()

    if launch_command_world(args.gpus, os.environ)[1]:  # --gpus N without a launcher: start the N ranks here
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)")
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
    from parcels_amd.distributed import shard_slice
    from tests.case_utils import build_fieldset, build_pset

    t_start_wall = time.perf_counter()
    legs = {}  # wall seconds of every leg of this run (rank 0): where the minutes of a default run go
    npart = int(args.particles)
    K, W = args.steps, args.warmup
    shard = shard_slice(world * npart, rank, world)  # one id space, sharded by id
    case = c2_case(seed=1, lo=shard.start, hi=shard.stop)
    nt = len(case["time_s"])
    if (K + W) * case["dt"] > case["time_s"][-1]:
        raise SystemExit(f"steps+warmup must stay within the {nt}-level time interval")
    fs = build_fieldset(case)
    fs.to_device(device=local_rank)
    pset = build_pset(case, fs, sort_by_cell=bool(args.sort))
    pset._data["particle_id"] += shard.start
    kern = pa.Kernel([pa.AdvectionRK4], pset)
    eng = fs._engine_or_create()
    # (the shader-clock probe -- 16 wavefronts on a second stream spinning 1 ms beside an advection kernel, pk_exec_stats.sclk_mhz -- runs on
    # ONE extra, untimed repetition behind the timed ones: nothing spins beside a timed launch)
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

    legs["setup_fieldset_particles_h2d"] = time.perf_counter() - t_start_wall
    # measured device-to-device copy bandwidth (float4 copy of 1 GiB): the practical HBM ceiling, and the calibration dispatch of
    # the FETCH_SIZE / WRITE_SIZE counter passes (tools/pmc_summary.py)
    copy_gbps = eng.ctx.copy_bandwidth(1 << 30, 5) if rank == 0 else 0.0
    try:
        di = eng.ctx.device_info()
        device = {"name": di["name"], "arch": di["arch"], "compute_units": di["compute_units"], "clock_mhz": di["clock_khz"] / 1e3,
                  "hbm_gb": di["total_mem"] / 1e9}
    except Exception:
        device = None

    # warmup: W steps (also pays the one-off cell sort)
    sort_ms = 0.0
    if W > 0:
        st_w = eng.execute(kern.kernel_ids, endtime=W * dt, dt0=dt, sort_by_cell=int(args.sort), t_start=0.0)
        sort_ms = st_w["sort_ms"]
    # The timed region, repeated: every repetition restores the device columns to the state after the warm-up (pk_particles_checkpoint /
    # _restore, outside the timed region) and times EXACTLY K steps between two barriers.  One repetition of the headline is a 9 ms
    # launch: a single sample of it says little (clock ramp, first touch), the median of several is what is quoted.
    import ctypes as C

    eng.ctx.check(eng.lib.pk_particles_checkpoint(eng.ctx.handle), "pk_particles_checkpoint")
    reps = max(int(args.reps), 1)
    rep_el, rep_kms, rep_steps, rep_sclk = [], [], [], []
    settle_ms = None
    if reps > 1:  # one untimed run of the timed region first (the first launch after the sort runs on cold caches and a ramping clock)
        sync()
        t0 = time.perf_counter()
        eng.execute(kern.kernel_ids, endtime=(W + K) * dt, dt0=dt, sort_by_cell=0, t_start=W * dt)
        sync()
        settle_ms = (time.perf_counter() - t0) * 1e3
    for r in range(reps):
        if r > 0 or settle_ms is not None:
            eng.ctx.check(eng.lib.pk_particles_restore(eng.ctx.handle), "pk_particles_restore")
        sync()
        t0 = time.perf_counter()
        st = eng.execute(kern.kernel_ids, endtime=(W + K) * dt, dt0=dt, sort_by_cell=0, t_start=W * dt)
        sync()
        rep_el.append(time.perf_counter() - t0)
        rep_kms.append(st["kernel_ms"])
        rep_steps.append(float(st["steps"]))
    # one more repetition of the same launch, untimed, with the clock probe spinning beside it
    probe_kms = None
    try:
        eng.ctx.set_option("clock_probe", 1)
        eng.ctx.check(eng.lib.pk_particles_restore(eng.ctx.handle), "pk_particles_restore")
        sync()
        st_p = eng.execute(kern.kernel_ids, endtime=(W + K) * dt, dt0=dt, sort_by_cell=0, t_start=W * dt)
        sync()
        if st_p.get("sclk_mhz"):
            rep_sclk.append(float(st_p["sclk_mhz"]))
        probe_kms = float(st_p["kernel_ms"])
    finally:
        eng.ctx.set_option("clock_probe", 0)
    # the write-out exchange (not a step), straight from the device columns: the all-gather of the to-write columns that the north star
    # names, and the gather-to-rank-0 that ParticleFile.write uses (parcels_amd/distributed.py) -- both timed, neither in `value`
    t_ag = t_g0 = 0.0
    comm = {}
    if dist is not None:
        from parcels_amd.distributed import allgather_output, device_write_rows, gather_write_columns

        from parcels_amd.distributed import ensure_comm

        # the library's own RCCL communicator (include/parcels_hip.h: pk_comm_init); False in the gloo rehearsal.  Should it fail on ANY rank
        # (no librccl.so next to the library, an RCCL error in the first exchange), every rank falls back to the torch.distributed exchange
        # together -- the line must survive, and says which transport it measured
        cabi, cabi_err = False, None
        try:
            cabi = ensure_comm(eng)
            allgather_output(eng, world, fetch=False) if cabi else None  # (untimed first exchange: staging buffers are allocated on first use)
        except Exception as e:
            cabi, cabi_err = False, repr(e)[:500]
        okf = torch.tensor([0 if cabi_err else 1], dtype=torch.int64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if int(okf.item()) == 0:
            cabi = False
            os.environ["PARCELS_AMD_TORCH_EXCHANGE"] = "1"
            try:
                eng.comm_destroy()
            except Exception:
                pass
        sync()
        t1 = time.perf_counter()
        gathered = allgather_output(eng, world, fetch=False) if cabi else allgather_output(eng, world)
        sync()
        t_ag = time.perf_counter() - t1
        assert (int(gathered["counts"].sum()) if cabi else gathered["particle_id"].shape[0]) == world * npart
        del gathered
        names_w = ["particle_id", "t", "z", "y", "x"]
        if cabi:  # what ParticleFile.write calls: the write filter on the device, counts all-gathered, rows to rank 0 -- pk_gather_rows_to_root
            sync()
            t1 = time.perf_counter()
            eng.gather_rows(names_w, (W + K) * dt, fetch=False)
            sync()
            t_g0 = time.perf_counter() - t1
            assert int(eng.comm_last_counts.sum()) == world * npart, eng.comm_last_counts
        else:
            cols = device_write_rows(eng, names_w, (W + K) * dt)  # the product's write filter, on the device
            sync()
            t1 = time.perf_counter()
            rooted = gather_write_columns(cols, device=local_rank)  # (host tensors over gloo)
            sync()
            t_g0 = time.perf_counter() - t1
            if rank == 0:
                assert rooted["particle_id"].shape[0] == world * npart, rooted["particle_id"].shape
            del rooted, cols
        try:
            ver = torch.cuda.nccl.version()
        except Exception:
            ver = None
        comm = {"backend": dist.get_backend(), "n_ranks_seen": dist.get_world_size(), "rccl_version": ".".join(str(v) for v in ver) if ver else None,
                "exchange": "C ABI (pk_allgather_output / pk_gather_rows_to_root: RCCL opened by libparcels_hip.so)" if cabi else "torch.distributed",
                **({"c_abi_exchange_error": cabi_err} if cabi_err else {})}
    # what the lock-step points of a sharded ParticleSet cost (N > 1): four more steps through DeviceEngine.execute with the batch agreements
    # installed (parcels_amd.distributed.batch_agreement: after every pass the ranks all-reduce the first erring iteration / failing sample, at
    # the end the error codes) -- outside the timed region; calls and seconds inside the all-reduces, max over ranks
    agreements = None
    if dist is not None and (W + K + 4) * dt <= case["time_s"][-1]:
        try:
            from parcels_amd.distributed import batch_agreement

            eng.agree_min, eng.agree_codes = batch_agreement(None, local_rank, engine=eng)
            sync()
            t1 = time.perf_counter()
            st_a = eng.execute(kern.kernel_ids, endtime=(W + K + 4) * dt, dt0=dt, sort_by_cell=0, t_start=(W + K) * dt)
            sync()
            wall_a = time.perf_counter() - t1
            ag = dict(eng.agree_min.stats)
            agreements = [float(ag["calls"]), float(ag["seconds"]), wall_a, float(st_a["kernel_ms"])]
        except Exception as e:  # the headline line must survive this leg
            agreements = None
            print(f"[bench] rank {rank}: agreement leg failed: {e!r}", file=sys.stderr)
        finally:
            eng.agree_min = eng.agree_codes = None
    elt = torch.tensor(rep_el, device="cuda", dtype=torch.float64)  # per repetition: the MAX over ranks ...
    steps_t = torch.tensor([rep_steps[-1]], device="cuda", dtype=torch.float64)
    kmst = torch.tensor(rep_kms, device="cuda", dtype=torch.float64)
    kmin = kmst.clone()
    if dist is not None:
        dist.all_reduce(elt, op=dist.ReduceOp.MAX)
        dist.all_reduce(steps_t, op=dist.ReduceOp.SUM)
        dist.all_reduce(kmst, op=dist.ReduceOp.MAX)
        dist.all_reduce(kmin, op=dist.ReduceOp.MIN)
    rep_el = [float(v) for v in elt.tolist()]
    rep_kms_max = [float(v) for v in kmst.tolist()]
    rep_kms_min = [float(v) for v in kmin.tolist()]
    order = sorted(range(reps), key=lambda i: rep_el[i])
    med = order[(reps - 1) // 2]  # ... and of those the median repetition (the lower one of an even count: an actual run, not an average)
    el = rep_el[med]
    kms = torch.tensor([rep_kms_max[med]], dtype=torch.float64)
    total_steps = float(steps_t.item())
    agt = torch.tensor([t_ag, t_g0] + (agreements or [-1.0, -1.0, -1.0, -1.0]), device="cuda", dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(agt, op=dist.ReduceOp.MAX)
    t_ag, t_g0 = (float(v) for v in agt.tolist()[:2])
    ag_max = [float(v) for v in agt.tolist()[2:]]
    t_d2h = time.perf_counter()
    eng.d2h()
    t_d2h = time.perf_counter() - t_d2h
    ok = bool(np.all(pset._data["state"] == pa.StatusCode.EndofLoop))
    # TEST INFRASTRUCTURE inside the measurement script (like `secondary[*].check`): the first ids of the timed run, re-run ALONE through the
    # CPU oracle over the same W + K steps -- particles are independent, so the subset must reproduce: state, ei, t, ids exactly, positions to
    # 1e-12 relative.  After the timed region; never part of `value`.
    headline_check = None
    if rank == 0 and args.check:
        try:
            headline_check = check_headline(case, pset._data, int(args.check), (W + K) * dt)
        except AssertionError as e:
            headline_check = {"passed": False, "error": str(e)[:2000]}
    if rank == 0:
        value = total_steps / el
        kernel_s = float(kms.item()) * 1e-3
        per_gpu_steps = total_steps / world
        algo_gbps = ALGO_BYTES_PER_STEP_C2_RK4 * per_gpu_steps / kernel_s / 1e9 if kernel_s > 0 else 0.0
        peak_cycles = N_SIMD * PEAK_CLOCK_GHZ  # G SIMD-cycles per second
        evals_s = per_gpu_steps * 4 / kernel_s if kernel_s > 0 else 0.0  # RK4: 4 evaluations per particle-step; per lane
        algo_tflops = ALGO_FLOPS_PER_EVAL_C2 * evals_s / 1e12
        # `frac` is a fraction of a ROOF: algorithmic fp64 flops over the launch time against the fp64 vector peak (the kernel is bound by
        # fp64-rate VALU issue, its gathers are served by L2 / Infinity Cache).  `valu_busy_frac` (rounds 1-4 quoted it as `frac`) is the share
        # of SIMD cycles with a VALU instruction in flight -- utilisation, not a roof: moves, selects and address arithmetic count in it.
        roof = {"bound": "valu_fp64", "achieved": algo_tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": algo_tflops / FP64_VECTOR_PEAK_TFLOPS,
                # what `frac` is, spelled out so that trend lines across rounds compare like with like: rounds 1-4 printed the
                # VALU-busy share there (now `valu_busy_frac`), rounds 5-6 ALGORITHMIC flops over the fp64 vector peak (= `algorithmic_frac`);
                # the flops this binary really executes are `frac_executed_fp64` (PMC instruction count x static fp64 mix, below)
                "frac_definition": "algorithmic fp64 flops (143.46 per velocity evaluation, an FMA = 2) / kernel time / 78.6 TFLOP/s",
                "algorithmic_frac": algo_tflops / FP64_VECTOR_PEAK_TFLOPS,
                "traffic": None, "algorithmic_fp64_flops_per_evaluation": ALGO_FLOPS_PER_EVAL_C2,
                "frac_no_fma_peak": algo_tflops / (FP64_VECTOR_PEAK_TFLOPS / 2) * (121.59 / ALGO_FLOPS_PER_EVAL_C2),  # one op per lane and slot: NumPy never fuses (-ffp-contract=off)
                "kernel": "pk::advect_fast_kernel<double, 0, false> (csrc/pk_kernels.h, pk_fast_agrid.h)", "kernel_ms_per_launch": float(kms.item()),
                # shader clock DURING the timed launches (a cycle-counter / 100 MHz-counter probe spinning beside each kernel for 1 ms; median over
                # the first millisecond of it): reconciles this line with a trace taken at another clock
                "sclk_mhz": (rep_sclk[0] if rep_sclk else None),  # (of the untimed probe repetition that follows the timed ones; its kernel ms beside it)
                "sclk_probe_rep_kernel_ms": probe_kms,
                "hbm": None,
                "algorithmic": {"note": "SURVEY 8(d) byte model; these bytes are served by L2 / Infinity Cache, this is NOT an HBM fraction",
                                "bytes_per_particle_step": ALGO_BYTES_PER_STEP_C2_RK4, "bytes_per_launch": ALGO_BYTES_PER_STEP_C2_RK4 * per_gpu_steps,
                                "achieved_gbps": algo_gbps, "over_hbm_peak": algo_gbps / HBM_PEAK_GBPS},
                "measured_copy_gbps": copy_gbps}
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):  # per-particle-step counters of the rocprofv3 PMC passes (tools/pmc_summary.py), scaled to THIS launch
            try:
                pj = json.load(open(pmc))
                pp = pj["per_particle_step"]
                busy = pp["valu_busy_simd_cycles"] * per_gpu_steps  # SIMD-cycles
                roof["valu_busy_frac"] = busy / kernel_s / 1e9 / peak_cycles  # at the 2.4 GHz peak clock
                roof["valu_insts_per_wave_evaluation"] = pp["valu_insts_per_wave_eval"]
                if pp.get("fetch_bytes") is not None and pp.get("write_bytes") is not None:
                    traffic = (pp["fetch_bytes"] + pp["write_bytes"]) * per_gpu_steps
                    roof["traffic"] = traffic
                    roof["hbm"] = {"achieved": traffic / kernel_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                   "frac": traffic / kernel_s / 1e9 / HBM_PEAK_GBPS, "bytes_per_particle_step": pp["fetch_bytes"] + pp["write_bytes"]}
                roof["counters_source"] = pj.get("source")
                roof["counters_stale"] = counters_stale("AdvectionRK4", pj)  # the kernel changed since the PMC passes
                isa = os.path.join(ROOT, "profiles", "isa_latest.json")
                if os.path.exists(isa):
                    # what the busy share is made of (tools/isa_histogram.py: instruction classes of the kernel's inner loops from hipcc -S):
                    # EXECUTED fp64 arithmetic of this binary, next to the algorithmic figure `frac` is built on
                    ij = json.load(open(isa))
                    ops = pp["valu_insts_per_wave_eval"] * ij["fp64_arith_fraction_of_valu"]
                    flops = pp["valu_insts_per_wave_eval"] * ij["fp64_flops_per_valu_instruction_per_lane"]
                    roof["executed_fp64_ops_per_evaluation"] = ops  # fp64 arithmetic wave-instructions per evaluation
                    roof["executed_fp64_flops_per_evaluation"] = flops  # per lane, an FMA counts 2
                    roof["frac_executed_fp64"] = flops * evals_s / (FP64_VECTOR_PEAK_TFLOPS * 1e12)
                    roof["valu_class_fractions"] = ij["valu_class_fractions"]
                    roof["valu_cycles_per_instruction_model"] = ij["valu_cycles_per_instruction"]  # fp64 / conversions 4 cycles, everything else 2
                    roof["frac_issue_model"] = pp["valu_insts_per_wave_eval"] * ij["valu_cycles_per_instruction"] * (evals_s / 64) / (peak_cycles * 1e9)
                    roof["isa_source"] = "profiles/isa_latest.json (tools/isa_histogram.py, static mix of the inner loops x the PMC instruction count)"
            except Exception as e:  # a malformed summary must not kill the bench line
                roof["counters_error"] = repr(e)
        t_all = el + t_h2d + t_d2h + sort_ms * 1e-3
        out = {
            "metric": "RK4 particle-steps/sec",
            "value": value,
            "unit": "particle-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": el / K * 1e3,
            # the timed region was repeated: `value`, `ms_per_step`, `roofline` are those of the median repetition
            "timed_reps": {"n": reps, "statistic": "median", "wall_ms": {"min": min(rep_el) * 1e3, "median": el * 1e3, "max": max(rep_el) * 1e3},
                           "kernel_ms": {"min": min(rep_kms_max), "median": sorted(rep_kms_max)[(reps - 1) // 2], "max": max(rep_kms_max), "n": reps},
                           "spread": (max(rep_el) - min(rep_el)) / el, "untimed_settling_rep_wall_ms": settle_ms,
                           **({"kernel_ms_slowest_vs_fastest_rank_of_median_rep": [rep_kms_max[med], rep_kms_min[med]]} if world > 1 else {})},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C2: 3D rectilinear A-grid 360x180x50x24 fp64 U,V (spherical), AdvectionRK4, dt=3600 s",
                       "particles_per_gpu": npart, "particle_dtype": "f64", "cell_sorted": bool(args.sort),
                       "parallelism": f"one id space of {world * npart} particles sharded by id x{world}, fields replicated",
                       "all_states_endofloop": ok, **({"rehearsal_shared_gpu_gloo": True} if rehearsal else {})},
            "check": headline_check,
            "roofline": roof,
            # boundary costs outside the timed steps (rank 0): host<->device copies of the particle columns, the one-off cell sort
            "host_boundary": {"h2d_ms": t_h2d * 1e3, "d2h_ms": t_d2h * 1e3, "cell_sort_ms": sort_ms,
                              "value_pcie_inclusive": total_steps / (el + t_h2d + t_d2h)},
            "value_end_to_end": total_steps / t_all,  # H2D of the particle columns + cell sort + K steps + D2H
            "writeout_allgather_ms": t_ag * 1e3 if world > 1 else None,
            "writeout_gather_to_root_ms": t_g0 * 1e3 if world > 1 else None,  # what ParticleFile.write does (rows of the write filter -> rank 0)
            "comm": comm or None,
            # 4 more steps with the batch agreements of a sharded ParticleSet installed (outside `value`): all-reduce calls per rank, seconds
            # inside them / wall / kernel ms, each the max over ranks
            "batch_agreements": ({"steps": 4, "calls": int(ag_max[0]), "seconds_in_allreduce_max_rank": ag_max[1], "wall_ms_max_rank": ag_max[2] * 1e3,
                                  "kernel_ms_max_rank": ag_max[3]} if world > 1 and ag_max[0] >= 0 else None),
            "device": device,
            "value_incl_writeout": total_steps / (el + t_ag) if world > 1 else None,
        }
        ref = os.path.join(ROOT, "profiles", "r02_cpu_reference.json")
        if os.path.exists(ref):
            try:
                rj = json.load(open(ref))
                out["cpu_baseline_reference"] = {"value": rj["all_cores"]["value"], "unit": rj["unit"], "cores": rj["all_cores"]["processes"],
                                                 "kind": "reference", "single_process_value": rj["single_process"]["value"],
                                                 "sample": f"{rj['all_cores']['particles']} particles x {rj['all_cores']['steps']} steps, {rj['what']}",
                                                 "box": rj["box"], "off_box": True,
                                                 "source": "profiles/r02_cpu_reference.json (tools/time_reference_cpu.py): NOT this box -- the reference is Python and does not travel"}
            except Exception:
                pass
        legs["headline_warmup_and_timed_reps"] = time.perf_counter() - t_start_wall - legs["setup_fieldset_particles_h2d"]
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N = 1 only
            _t = time.perf_counter()
            out["cpu_baseline"] = cpu_baseline(case, steps=K, sample=min(args.cpu_sample, npart))
            legs["cpu_baseline"] = time.perf_counter() - _t
        if args.long_run and world == 1:
            _t = time.perf_counter()
            try:
                out["long_run"] = long_run(case, fs, int(args.long_run))
            except Exception as e:
                out["long_run"] = {"error": repr(e)[:2000]}
            legs["long_run"] = time.perf_counter() - _t
        if args.repeat_execute and world == 1:
            _t = time.perf_counter()
            try:
                out["repeat_execute"] = repeat_execute(case, fs, int(args.repeat_execute), K)
            except Exception as e:
                out["repeat_execute"] = {"error": repr(e)[:2000]}
            legs["repeat_execute"] = time.perf_counter() - _t
        if args.with_output and world == 1:
            _t = time.perf_counter()
            try:
                out["with_output"] = with_output(case, fs, int(args.with_output), 24)
            except Exception as e:
                out["with_output"] = {"error": repr(e)[:2000]}
            legs["with_output"] = time.perf_counter() - _t
        if args.secondary and world == 1:
            # release the headline's device and host memory first: C3 needs 36 GB of HBM and 48 GB of host arrays
            pset = kern = eng = fs = case = None
            import gc

            gc.collect()
            _t = time.perf_counter()
            try:
                out["secondary"] = secondary_runs(args)
            except Exception as e:  # the headline line must survive a failing secondary leg
                out["secondary"] = [{"error": repr(e)[:2000]}]
            legs["secondary"] = time.perf_counter() - _t
            if args.user_kernels:
                _t = time.perf_counter()
                try:
                    out["user_kernels"] = user_kernel_runs(int(args.user_kernels))
                except Exception as e:
                    out["user_kernels"] = {"error": repr(e)[:2000]}
                legs["user_kernels"] = time.perf_counter() - _t
    c4 = {}
    if args.c4:  # every rank takes part; rank 0 holds the result
        pset = kern = eng = fs = case = None  # (the headline's device and host memory: config 4 needs 36 GB of HBM per rank)
        import gc

        gc.collect()
        try:
            from tools import bench_configs as bc

            bc.run_c4(particles=args.c4, emit=c4.update)
        except Exception as e:
            c4 = {"error": repr(e)[:2000]}
    if rank == 0:
        if args.c4:
            out["c4"] = c4
        legs["total_after_imports"] = time.perf_counter() - t_start_wall
        out["legs_wall_s"] = legs
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
