#!/usr/bin/env python
"""Measure the non-headline BASELINE.json configurations on one MI355X (not the bench.py contract line):

  c3  3-D NEMO-like curvilinear C-grid nx=4322, ny=3059, nz=75, fp32 U,V,W, 1e7 particles, AdvectionRK4_3D,
      field-slab ring (nslots < nt) with asynchronous prefetch of the next level
  c4  the c3 grid with `--particles` per GPU on every rank of one node: launch with
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/bench_configs.py --config c4
      one id space sharded by id, fields replicated from ONE shared memory-mapped host copy, ParticleFile on rank 0 fed by the
      RCCL all-gather of the to-write columns
  c5  same grid, AdvectionRK45 (adaptive, divergent dt) and AdvectionDiffusionM1 (per-particle counter RNG)

`--scale s` shrinks nx, ny by s (default 1.0 = the BASELINE size).  Prints one JSON object per measured kernel list.
`--check N` re-runs the first N particle ids of every measured run through the CPU oracle on the same arrays (parity AT bench
size: deleted set, state, ei, t exact; positions 1e-12) and fails loudly on any difference.
Synthetic data: smooth analytic patterns on a smoothly warped lon/lat mesh.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def nemo_like_coords(nx, ny):
    """Node longitudes / latitudes of the synthetic curvilinear mesh (and the unit index ramps they are made of)."""
    i = np.arange(nx, dtype=np.float64)[None, :] / (nx - 1)
    j = np.arange(ny, dtype=np.float64)[:, None] / (ny - 1)
    lon = -170.0 + 340.0 * i + 2.0 * np.sin(2 * np.pi * j) * (0.3 + i) + 3.0 * j
    lat = -75.0 + 150.0 * j + 1.5 * np.sin(2 * np.pi * i) * (0.5 + 0.5 * j) - 2.0 * i
    return lon, lat, i, j


HASH_DIGESTS = os.path.join(ROOT, "tools", "bench_hash_digests.json")


def independent_hash_table(engine, lon, lat):
    """The spatial-hash table the bench-size CHECKER hands to the oracle must not simply be the thing checked.  The
    device-built table is used only after its SHA-256 digests (keys, starts, counts, faces, bit width, bounding box) were compared with
    the digest the HOST builder (parcels_amd/spatialhash.py, pinned to the reference's table by tests/test_spatialhash_reference.py)
    produced for this mesh -- committed in tools/bench_hash_digests.json by tools/make_bench_hash_digest.py (the host build of the
    1.3e7-face mesh takes minutes and ~20 GB of host memory: once, offline).  A mesh without a committed digest is built on the host here."""
    from parcels_amd import spatialhash as sh

    key = f"{lon.shape[1]}x{lon.shape[0]}"
    digests = json.load(open(HASH_DIGESTS)) if os.path.exists(HASH_DIGESTS) else {}
    if key in digests:
        table = engine.hash_table(0)
        got = sh.table_checksum(table)
        want = digests[key]["checksum"]
        bad = [k for k in want if (got[k] != want[k] if k != "bbox" else list(got[k]) != list(want[k]))]
        assert not bad, f"device-built hash table of the {key} mesh differs from the host builder's committed digest in {bad}"
        return table, {"source": "device table, SHA-256 equal to the host builder's committed digest", "digest_file": "tools/bench_hash_digests.json"}
    t0 = time.perf_counter()
    table = sh.SpatialHash(lon, lat, True).table()
    return table, {"source": "host builder (parcels_amd/spatialhash.py)", "build_s": time.perf_counter() - t0}


def nemo_like_dataset(nx, ny, nz, nt, with_kh=False, seed=0, shared_dir=None, generate=True):
    """shared_dir: U, V, W live in .npy files there and are memory-mapped (config c4: the ranks of one node share ONE copy of
    the field levels through the page cache instead of holding a private 48 GB each); generate=False opens them read-only."""
    import parcels_amd as pa

    t0 = time.perf_counter()
    lon, lat, i, j = nemo_like_coords(nx, ny)
    depth = np.concatenate([[0.0], np.cumsum(np.linspace(5.0, 150.0, nz - 1))])
    time_s = np.arange(nt) * 86400.0
    zprof = np.exp(-depth / 1500.0).astype(np.float32)[:, None, None]
    ii = i.astype(np.float32)
    jj = j.astype(np.float32)
    pu = (0.5 * np.cos(6 * np.pi * jj) * (1.0 + 0.3 * np.sin(10 * np.pi * ii))).astype(np.float32)
    pv = (0.3 * np.sin(8 * np.pi * ii) * np.cos(4 * np.pi * jj)).astype(np.float32)
    pw = (1e-4 * np.sin(12 * np.pi * ii) * np.sin(12 * np.pi * jj)).astype(np.float32)
    if shared_dir is None:
        U = np.empty((nt, nz, ny, nx), np.float32)
        V = np.empty((nt, nz, ny, nx), np.float32)
        W = np.empty((nt, nz, ny, nx), np.float32)
    else:
        os.makedirs(shared_dir, exist_ok=True)
        mode = "w+" if generate else "r"
        kw = dict(dtype=np.float32, shape=(nt, nz, ny, nx)) if generate else {}
        U, V, W = (np.lib.format.open_memmap(os.path.join(shared_dir, f"{c}.npy"), mode=mode, **kw) for c in "UVW")
    for k in range(nt if generate else 0):
        f = np.float32(1.0 + 0.2 * np.sin(2 * np.pi * k / max(nt, 2)))
        np.multiply(zprof, pu[None] * f, out=U[k])
        np.multiply(zprof, pv[None] * f, out=V[k])
        np.multiply(zprof, pw[None] * f, out=W[k])
    md = pa.SGrid2DMetadata(
        node_dimensions=("XG", "YG"), node_coordinates=("lon", "lat"),
        face_dimensions=(pa.FaceNodePadding("XC", "XG", pa.Padding.LOW), pa.FaceNodePadding("YC", "YG", pa.Padding.LOW)),
        vertical_dimensions=(pa.FaceNodePadding("ZC", "depth", pa.Padding.HIGH),),
    )
    dv = {"U": (("time", "ZC", "YC", "XG"), U), "V": (("time", "ZC", "YG", "XC"), V), "W": (("time", "depth", "YC", "XC"), W)}
    if with_kh:
        kh = (10.0 * (1.0 + 0.5 * np.tanh(3 * (2 * ii - 1))) * np.ones_like(jj)).astype(np.float32)
        dv["Kh_zonal"] = (("YG", "XG"), kh)
        dv["Kh_meridional"] = (("YG", "XG"), (10.0 * (1.0 + 0.3 * np.tanh(2 * (2 * jj - 1))) * np.ones_like(ii)).astype(np.float32))
    ds = pa.Dataset(dv, {"lon": (("YG", "XG"), lon), "lat": (("YG", "XG"), lat), "depth": (("depth",), depth),
                         "time": (("time",), time_s)}, sgrid=md)
    gen_s = time.perf_counter() - t0
    return ds, lon, lat, depth, gen_s


def seed_particles(lon, lat, depth, n, seed):
    rng = np.random.default_rng(seed)
    ny, nx = lon.shape
    ci = rng.uniform(0.1, 0.9, n) * (nx - 1)
    cj = rng.uniform(0.1, 0.9, n) * (ny - 1)
    i0, j0 = ci.astype(np.int64), cj.astype(np.int64)
    fi, fj = ci - i0, cj - j0

    def blend(a):
        return (a[j0, i0] * (1 - fi) * (1 - fj) + a[j0, i0 + 1] * fi * (1 - fj) + a[j0 + 1, i0] * (1 - fi) * fj + a[j0 + 1, i0 + 1] * fi * fj)

    return blend(lon), blend(lat), rng.uniform(5.0, 0.6 * depth[-1], n)


POS_BLOCK = 1 << 20


def seed_particles_block(lon, lat, depth, lo, hi, seed):
    """Particles lo..hi-1 of one global id space, drawn block-wise so that positions do not depend on the sharding."""
    xs, ys, zs = [], [], []
    for b in range(lo // POS_BLOCK, (max(hi, lo + 1) - 1) // POS_BLOCK + 1):
        x, y, z = seed_particles(lon, lat, depth, POS_BLOCK, seed=[seed, b])
        sl = slice(max(lo - b * POS_BLOCK, 0), min(hi - b * POS_BLOCK, POS_BLOCK))
        xs.append(x[sl]); ys.append(y[sl]); zs.append(z[sl])
    return np.concatenate(xs), np.concatenate(ys), np.concatenate(zs)


def run_c4(scale=1.0, particles=1e7, steps=24, nt=4, nslots=3, nz=75, output_every=6, verify_single=False, dt=3600.0, shared_dir=None, emit=None):
    """BASELINE config 4: the C3 grid, `particles` per GPU over all ranks of one node (launch with torch.distributed.run, one
    process per GPU), ONE id space sharded by id, fields replicated in HBM from one shared memory-mapped copy on the host, no
    collective on the data path, and a ParticleFile on rank 0 fed by the RCCL all-gather of the to-write columns every
    `output_every` steps.  verify_single: rank 0 re-runs the whole id space alone and compares the Parquet files byte for byte."""
    import torch
    import torch.distributed as dist

    import parcels_amd as pa
    from parcels_amd.distributed import shard_slice

    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    rehearsal = os.environ.get("PARCELS_AMD_BENCH_REHEARSAL") == "1"  # 1-GPU box: all ranks on cuda:0, gloo instead of RCCL
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    own_group = world > 1 and not dist.is_initialized()  # (bench.py --c4 calls this inside its own process group)
    if own_group:
        dist.init_process_group("gloo" if rehearsal else "nccl", **({} if rehearsal else {"device_id": torch.device("cuda", local_rank)}))
    shared_dir = shared_dir or os.environ.get("PK_C4_DIR", f"/dev/shm/pk_c4_{os.environ.get('MASTER_PORT', '0')}")
    nx, ny = max(int(4322 * scale), 32), max(int(3059 * scale), 32)
    n = int(particles)
    t0 = time.perf_counter()
    if rank == 0:
        nemo_like_dataset(nx, ny, nz, nt, shared_dir=shared_dir, generate=True)
    if world > 1:
        dist.barrier()
    ds, lon, lat, depth, _ = nemo_like_dataset(nx, ny, nz, nt, shared_dir=shared_dir, generate=False)
    gen_s = time.perf_counter() - t0
    out_path = os.path.join(shared_dir, "c4.parquet")

    def one_run(lo, hi, path, distributed):
        fs = pa.FieldSet.from_sgrid_conventions(ds, mesh="spherical", skip_field_data_validation=True)
        fs.to_device(local_rank, nslots=nslots)
        x, y, z = seed_particles_block(lon, lat, depth, lo, hi, seed=3)
        pset = pa.ParticleSet(fs, pclass=pa.get_default_particle(np.float64), x=x, y=y, z=z, t=np.zeros(hi - lo), particle_ids=np.arange(lo, hi),
                              sort_by_cell=True)
        pset.populate_indices()
        pf = pa.ParticleFile(path, outputdt=float(output_every * dt), mode="w", distributed=None if distributed else False)
        torch.cuda.synchronize()
        if distributed and world > 1:
            dist.barrier()
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t1 = time.perf_counter()
            pset.execute([pa.AdvectionRK4_3D, pa.DeleteParticle], dt=dt, runtime=steps * dt, output_file=pf)
            torch.cuda.synchronize()
            if distributed and world > 1:
                dist.barrier()
            wall = time.perf_counter() - t1
        return pset, pf, wall

    sl = shard_slice(world * n, rank, world)
    pset, pf, wall = one_run(sl.start, sl.stop, out_path, True)
    st = pset._last_stats
    ag = getattr(pset, "_agreement_stats", None) or {}  # the per-pass batch agreements of a collective run (parcels_amd.distributed.batch_agreement)
    vals = torch.tensor([wall, float(len(pset)), pf.gather_seconds, float(ag.get("seconds", 0.0))], dtype=torch.float64)
    if world > 1:
        mx = vals.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = vals.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        wall, remaining, gather_s, agree_s = float(mx[0]), int(sm[1]), float(mx[2]), float(mx[3])
    else:
        remaining, gather_s, agree_s = len(pset), pf.gather_seconds, float(ag.get("seconds", 0.0))
    if rank == 0:
        import pyarrow.parquet as pq

        rows = pq.ParquetFile(out_path).metadata.num_rows
        out = {"config": "c4", "kernels": "AdvectionRK4_3D", "grid": [nx, ny, nz, nt], "nslots": nslots, "n_gpus": world, "particles_per_gpu": n,
               "particles_total": world * n, "steps": steps, "output_every_steps": output_every, "wall_s_incl_writeout": wall,
               "particle_steps_per_s_wall": world * n * steps / wall, "writeout_gather_s_max_rank": gather_s, "parquet_rows": rows,
               "remaining_particles": remaining, "last_interval_kernel_ms_rank0": st["kernel_ms"], "dataset_generation_s": gen_s,
               "batch_agreements": {"calls_rank0": int(ag.get("calls", 0)), "seconds_max_rank": agree_s,
                                    "note": "all-reduces that make the shards one batch for the error stop / call-wide time error: inside wall_s_incl_writeout"},
               "shared_fields": shared_dir, **({"rehearsal_shared_gpu_gloo": True} if rehearsal else {})}
        if verify_single:
            single = os.path.join(shared_dir, "c4_single.parquet")
            one_run(0, world * n, single, False)
            out["byte_identical_to_single_process_file"] = open(single, "rb").read() == open(out_path, "rb").read()
            assert out["byte_identical_to_single_process_file"], "the gathered Parquet differs from the single-process file"
        if emit is None:
            print(json.dumps(out), flush=True)
        else:
            emit(out)
    if world > 1:
        dist.barrier()
        if own_group:
            dist.destroy_process_group()
    if rank == 0 and os.environ.get("PK_C4_KEEP") != "1":
        import shutil

        shutil.rmtree(shared_dir, ignore_errors=True)


def check_against_oracle(*, n_check, dsinfo, engine, pset, kernel_names, context, x, y, z, dt, runtime, nthreads=None):
    """TEST INFRASTRUCTURE inside the measurement script: the first `n_check` particle ids of the run just finished are re-run
    alone through the CPU oracle (oracle/parcels_oracle.c) on the SAME arrays and hash table -- particles are independent, so
    the subset must reproduce: deleted set, state, ei, t exactly; positions to 1e-12 of the coordinate scale (1e-11 with the Box-Muller
    draws of the stochastic kernels).  Raises AssertionError otherwise; returns the worst relative differences."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from case_utils import compare
    from oracle import c_oracle as co

    ds, lon, lat, depth = dsinfo
    fields, dims = {}, {}
    for name, (fdims, arr) in ds.items():
        if arr.ndim == 2:  # 2-D Kh fields: (YG, XG) -> (mockT, mockZ, YG, XG)
            arr, fdims = arr[None, None], ("mockT", "mockZ") + tuple(fdims)
        fields[name], dims[name] = arr, tuple(fdims)
    case = dict(name="bench_check", mesh="spherical", lon=lon, lat=lat, depth=depth, x_pad="low", y_pad="low", z_pad="high",
                time_s=np.arange(fields["U"].shape[0]) * 86400.0, fields=fields, field_dims=dims, cgrid=True, kernels=list(kernel_names),
                spatial_dtype="float64", x=x[:n_check], y=y[:n_check], z=z[:n_check], t0=None, dt=dt, runtime=runtime, seed=0,
                context={k: v for k, v in context.items() if k == "dres"}, populate=True)
    case["hash_table"], hash_source = independent_hash_table(engine, np.asarray(lon), np.asarray(lat))
    t0 = time.perf_counter()
    ref, err, _ = co.run_case(case, nthreads=nthreads or (os.cpu_count() or 1))
    oracle_s = time.perf_counter() - t0
    assert err is None, f"oracle raised {err}"
    sel = pset._data["particle_id"] < n_check
    got = {k: np.asarray(v)[sel] for k, v in pset._data.items()}
    stochastic = any(k.startswith("AdvectionDiffusion") for k in kernel_names)
    rtol = 1e-11 if stochastic else 1e-12
    # relative to the coordinate scale: longitudes run through 0 on this mesh, |x| itself is no yardstick there (tests/test_gpu_fuzz.py)
    scale = float(max(np.abs(lon).max(), np.abs(lat).max()))
    rep = compare(got, ref, rtol=rtol, atol_pos=rtol * scale, check_state="all", label="bench-size subset vs oracle",
                  skip=("dt",) if "AdvectionRK45" in kernel_names else ())
    worst_abs = {k: float(np.max(np.abs(np.asarray(got[k], dtype=np.float64) - np.asarray(ref[k], dtype=np.float64)))) if len(ref[k]) else 0.0
                 for k in ("x", "y", "z")}
    return {"n_check": int(n_check), "survivors": int(sel.sum()), "deleted": int(n_check - sel.sum()), "max_rel_diff": rep,
            "max_abs_diff": worst_abs, "tolerance": f"|a-b| <= {rtol:g} * (|b| + {scale:.0f})",
            "exact": ["particle ids of the survivors (= deleted set)", "state", "ei", "t"], "oracle_s": oracle_s, "oracle_hash_table": hash_source}


def run_config(config="c3", scale=1.0, particles=1e7, steps=24, nt=4, nslots=3, nz=75, hash="device", check=0, emit=print, dt=3600.0, reps=5,
               pairs_leg=True, only=None):
    """pairs_leg (c5): after the default runs (the 2-D kernels read the level rings), the same launches once more with the OPT-IN cell-packed pair
    copies of the staggered velocity ("velocity_pairs"): their kernel time, the HIP-event time of packing one level pair (pk_exec_stats.pack_ms)
    and the sum -- at config 5 one pair serves one 24 h launch, so `kernel_plus_pack_ms` is what a level of model time costs with them.
    reps: after one COLD run of a kernel list (the first launch after the FieldSet was built: first touch of the tables, clock ramp) the
    same run is repeated `reps` times from fresh ParticleSets of the same positions; `kernel_ms` is the MEDIAN of those, the cold one and
    min / max are reported next to it (`kernel_ms_stats`).  reps = 0: the cold run alone, like rounds 1-3."""
    import parcels_amd as pa

    nx, ny = max(int(4322 * scale), 32), max(int(3059 * scale), 32)
    n = int(particles)
    ds, lon, lat, depth, gen_s = nemo_like_dataset(nx, ny, nz, nt, with_kh=(config == "c5"))
    t0 = time.perf_counter()
    fs = pa.FieldSet.from_sgrid_conventions(ds, mesh="spherical", skip_field_data_validation=True)
    if config == "c5":
        fs.add_context("dres", 0.01)
    hash_s = 0.0
    if hash == "host":
        fs.gridset[0].get_spatial_hash()
        hash_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    fs.to_device(0, nslots=nslots)  # builds the spatial hash on the device unless the host table exists
    upload_s = time.perf_counter() - t0
    fs._engine.ctx.set_option("clock_probe", 1)  # pk_exec_stats.sclk_mhz: the shader clock while a kernel runs (16 wavefronts spin 1 ms beside it)
    x, y, z = seed_particles(lon, lat, depth, n, seed=3)
    runs = [("AdvectionRK4_3D", [pa.AdvectionRK4_3D, pa.DeleteParticle], None)] if config == "c3" else [
        ("AdvectionRK45", [pa.AdvectionRK45, pa.DeleteParticle], "rk45"),
        ("AdvectionDiffusionM1", [pa.AdvectionDiffusionM1, pa.DeleteParticle], "m1"),
    ]
    if only:  # (A/B runs: one kernel list of config 5)
        runs = [r for r in runs if r[2] == only]
    results = []
    pack_per_pair = None  # (the copies packed for the first 2-D kernel list serve the next one: one measurement of the packing)
    for label, kernels, kind in runs:
        pclass = pa.get_default_particle(np.float64)
        if kind == "rk45":
            pclass = pclass.add_variable(pa.Variable("next_dt", dtype=np.float64, initial=dt))
        import warnings

        kms_all, sclk_all = [], []
        for r in range(1 + max(int(reps), 0)):
            # RK45 mode is keyed on the context (kernel.py:118): do not leak it into the other runs -- and every Kernel construction
            # divides RK45_tol by deg2m again (kernel.py:144-145, reproduced), so a repetition starts from the defaults as well
            for key in ("RK45_tol", "RK45_min_dt", "RK45_max_dt"):
                fs.context.pop(key, None)
            pset = pa.ParticleSet(fs, pclass=pclass, x=x, y=y, z=z, t=np.zeros(n), sort_by_cell=True)
            pset.populate_indices()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                t0 = time.perf_counter()
                pset.execute(kernels, dt=dt, runtime=steps * dt)
                wall = time.perf_counter() - t0
            kms_all.append(float(pset._last_stats["kernel_ms"]))
            sclk_all.append(float(pset._last_stats.get("sclk_mhz") or 0.0))
        st = dict(pset._last_stats)
        timed = sorted(kms_all[1:]) or kms_all
        st["kernel_ms"] = timed[(len(timed) - 1) // 2]
        out = {
            "config": config, "kernels": label, "grid": [nx, ny, nz, nt], "nslots": nslots, "particles": n,
            "steps_requested": steps, "particle_steps": int(st["steps"]), "attempts": int(st["attempts"]),
            "kernel_ms": st["kernel_ms"], "sort_ms": st["sort_ms"], "launches": st["launches"],
            "sclk_mhz": (sclk_all[kms_all.index(st["kernel_ms"])] or None), "sclk_mhz_all": sclk_all,  # shader clock beside the median launch / all launches (cold first)
            "kernel_ms_stats": {"cold": kms_all[0], "min": timed[0], "median": st["kernel_ms"], "max": timed[-1], "n": len(timed), "statistic": "median of the timed launches" if len(kms_all) > 1 else "the cold launch"},
            "particle_steps_per_s_kernel": st["steps"] / (st["kernel_ms"] * 1e-3) if st["kernel_ms"] else None,
            "particle_steps_per_s_wall_incl_h2d_d2h": st["steps"] / wall, "wall_s": wall,
            "remaining_particles": len(pset), "state_counts": st["state_counts"],
            "stream_host_s": {k: st.get(k) for k in ("commit_s", "prefetch_s", "wait_s")},
            "upload_stats": fs._engine.ctx.upload_stats() if getattr(fs, "_engine", None) is not None else None,
            "dataset_generation_s": gen_s, "hash_build": hash, "host_hash_build_s": hash_s, "device_create_s": upload_s,
        }
        if config == "c5" and pairs_leg:
            eng = fs._engine
            eng.ctx.set_option("velocity_pairs", 1)
            pk, packs, pack_ms = [], 0, 0.0
            try:
                for r in range(3):
                    for key in ("RK45_tol", "RK45_min_dt", "RK45_max_dt"):
                        fs.context.pop(key, None)
                    p2 = pa.ParticleSet(fs, pclass=pclass, x=x, y=y, z=z, t=np.zeros(n), sort_by_cell=True)
                    p2.populate_indices()
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        p2.execute(kernels, dt=dt, runtime=steps * dt)
                    s2 = p2._last_stats
                    pk.append(float(s2["kernel_ms"]))
                    if s2.get("packs"):
                        packs, pack_ms = int(s2["packs"]), float(s2["pack_ms"])
                    del p2
            finally:
                eng.ctx.set_option("velocity_pairs", 0)
            kp = sorted(pk[1:])[(len(pk[1:]) - 1) // 2]
            if packs:
                pack_per_pair = pack_ms / packs
            per_pair = pack_per_pair
            out["velocity_pairs"] = {
                "default": "off", "kernel_ms": kp, "kernel_ms_all": pk, "packs_first_launch": packs, "pack_ms_first_launch": pack_ms,
                "pack_ms_per_pair": per_pair, "kernel_plus_pack_ms": (kp + per_pair) if per_pair is not None else None,
                "note": "one level pair serves the 24 h of model time this launch covers: kernel + pack is the cost per level WITH the copies, "
                        "`kernel_ms` of the parent object the cost WITHOUT them"}
        if check:
            out["check"] = check_against_oracle(n_check=min(int(check), n), dsinfo=({k: (v.dims, v.data) for k, v in ds.data_vars.items()}, lon, lat, depth), engine=fs._engine,
                                                pset=pset, kernel_names=[label, "DeleteParticle"], context=fs.context, x=x, y=y, z=z, dt=dt,
                                                runtime=steps * dt)
        emit(json.dumps(out), flush=True) if emit is print else emit(out)
        results.append(out)
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=["c3", "c4", "c5"])
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--particles", type=float, default=1e7)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--nt", type=int, default=4)
    ap.add_argument("--nslots", type=int, default=3)
    ap.add_argument("--nz", type=int, default=75)
    ap.add_argument("--hash", default="device", choices=["device", "host"], help="where the Morton table of the grid is built")
    ap.add_argument("--dt", type=float, default=3600.0)
    ap.add_argument("--check", type=float, default=0, help="re-run the first N particles through the CPU oracle and compare (0 = off)")
    ap.add_argument("--reps", type=int, default=5, help="c3 / c5: timed repetitions after the cold run (kernel_ms = their median)")
    ap.add_argument("--output-every", type=int, default=6, help="c4: steps between write-outs")
    ap.add_argument("--verify-single", action="store_true", help="c4: rank 0 re-runs the whole id space alone and compares the files")
    ap.add_argument("--only", default=None, choices=["rk45", "m1"], help="c5: run only this kernel list")
    ap.add_argument("--pairs-leg", type=int, default=1, help="c5: also time the opt-in pair copies (kernel + pack per level pair)")
    a = ap.parse_args()
    if a.config == "c4":
        run_c4(a.scale, a.particles, a.steps, a.nt, a.nslots, a.nz, a.output_every, a.verify_single, a.dt)
        return
    run_config(a.config, a.scale, a.particles, a.steps, a.nt, a.nslots, a.nz, a.hash, int(a.check), dt=a.dt, reps=a.reps, pairs_leg=bool(a.pairs_leg), only=a.only)


if __name__ == "__main__":
    main()
