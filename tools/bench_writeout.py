#!/usr/bin/env python
"""Write-out off the critical path: the C2 workload of bench.py with a ParticleFile written every `--every` steps (a comma list: a sweep
over cadences), per cadence once without output, once with the inline path (D2H of the to-write columns + write filter + Parquet encode
between two launches) and once with the asynchronous path (write filter on the device -> pinned snapshot on the copy stream -> writer
thread), same file byte for byte.  Per cadence: wall seconds, the write-out cost of both paths, `output_hidden_frac` = the share of the inline
cost that the asynchronous path took out of the wall clock, and where the writer's time went.  Then the Parquet encode of ONE table on its
own: this writer with one page per column chunk (round 5) and with 512K- / 128K-row pages, by thread count, against pyarrow's.  One JSON object."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=float, default=1e7)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--every", default="2,24", help="steps between tables; a comma list sweeps")
    ap.add_argument("--compression", default="zstd")
    a = ap.parse_args()
    import torch

    import parcels_amd as pa
    from bench import c2_case
    from tests.case_utils import build_fieldset, build_pset

    n = int(a.particles)
    case = c2_case(seed=1, lo=0, hi=n)
    fs = build_fieldset(case)
    fs.to_device()
    comp = None if a.compression in ("none", "None") else a.compression
    out = {"workload": f"C2, {n} fp64 particles, AdvectionRK4, {a.steps} steps, ParticleFile (compression {comp}) at the cadences below", "cadences": []}
    row_bytes = 40
    with tempfile.TemporaryDirectory() as tmp:
        def run(mode, every, steps):
            pset = build_pset(case, fs, sort_by_cell=True, resort_every=0)
            pset.async_output = mode == "async"
            path = os.path.join(tmp, f"{mode}_{every}.parquet")
            pf = None if mode == "none" else pa.ParticleFile(path, outputdt=float(every * case["dt"]), compression=comp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pset.execute([pa.AdvectionRK4, pa.DeleteParticle], dt=case["dt"], runtime=steps * case["dt"], output_file=pf)  # (DeleteParticle: over weeks a few particles reach the edge)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            r = {"wall_s": wall}
            if pf is not None:
                r["file_MB"] = os.path.getsize(path) / 1e6
                r["parquet_writer_s"] = getattr(pf, "writer_seconds", None)
                r["sha"] = __import__("hashlib").sha256(open(path, "rb").read()).hexdigest()[:16]
                os.remove(path)
            if mode == "async":
                r["writer"] = getattr(pset, "_last_writer_stats", None)
            return r, pset

        run("async", 2, 4)  # warm-up: pins the snapshot buffers, starts the pools
        run("inline", 2, 4)
        for every in [int(v) for v in str(a.every).split(",")]:
            res = {}
            for mode in ("none", "inline", "async"):
                res[mode], pset = run(mode, every, a.steps)
            tables = a.steps // every + 1
            ci, ca = res["inline"]["wall_s"] - res["none"]["wall_s"], res["async"]["wall_s"] - res["none"]["wall_s"]
            out["cadences"].append({
                "every_steps": every, "tables": tables, "wall_s": {k: v["wall_s"] for k, v in res.items()}, "write_out_cost_s": {"inline": ci, "async": ca},
                "per_table_ms": {"inline": ci / tables * 1e3, "async": ca / tables * 1e3},
                "async_not_slower_than_inline": bool(res["async"]["wall_s"] <= res["inline"]["wall_s"] * 1.02),
                "output_hidden_frac": 1.0 - ca / ci if ci > 0 else None, "table_GB_per_s": {"inline": tables * n * row_bytes / 1e9 / ci, "async": tables * n * row_bytes / 1e9 / max(ca, 1e-9)},
                "byte_identical": res["inline"]["sha"] == res["async"]["sha"], "file_MB": res["async"]["file_MB"],
                "async_writer": res["async"].get("writer"), "parquet_writer_s": {"inline": res["inline"].get("parquet_writer_s"), "async": res["async"].get("parquet_writer_s")}})
        # the Parquet encode of ONE table of n rows (the default Variables: particle_id, t, z, y, x)
        import pyarrow as pyarrow
        import pyarrow.parquet as pq

        from parcels_amd.parquet_writer import FastParquetWriter

        cols = {k: np.ascontiguousarray(pset._data[k]) for k in ("particle_id", "t", "z", "y", "x")}
        schema = pyarrow.schema([pyarrow.field(k, pyarrow.from_numpy_dtype(v.dtype)) for k, v in cols.items()])
        enc = {}
        k = 0
        for label, kw in (("one_page_per_chunk_8_threads (round 5)", dict(threads=8, page_rows=1 << 20)), ("one_page_per_chunk_32_threads (round 5)", dict(threads=32, page_rows=1 << 20)),
                          ("pages_512k_rows_8_threads", dict(threads=8)), ("pages_512k_rows_32_threads", dict(threads=32)), ("pages_512k_rows_64_threads (default)", dict(threads=64)),
                          ("pages_128k_rows_64_threads", dict(threads=64, page_rows=1 << 17)), ("pages_512k_rows_64_threads_uncompressed", dict(threads=64, compression=None))):
            best, sec = 1e9, None
            for _ in range(3):
                k += 1
                kw2 = dict(kw)
                c2 = kw2.pop("compression", comp)
                w = FastParquetWriter(os.path.join(tmp, f"enc{k}.parquet"), schema, compression=c2, **kw2)  # (a new path each time: no truncation of an old file in the timing)
                t0 = time.perf_counter()
                w.write_columns(cols)
                el = time.perf_counter() - t0
                w.close()
                if el < best:
                    best, sec = el, dict(w.seconds)
                os.remove(os.path.join(tmp, f"enc{k}.parquet"))
            enc[label] = {"seconds": best, "GB_per_s": n * row_bytes / 1e9 / best, "waiting_for": sec}
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            with pq.ParquetWriter(os.path.join(tmp, "enc_pa.parquet"), schema, compression=comp or "none", use_dictionary=False) as w:
                w.write_table(pyarrow.table(cols, schema=schema))
            best = min(best, time.perf_counter() - t0)
        enc["pyarrow_writer"] = {"seconds": best, "GB_per_s": n * row_bytes / 1e9 / best}
        with FastParquetWriter(os.path.join(tmp, "enc.parquet"), schema, compression=comp) as w:
            w.write_columns(cols)
        enc["same_table_as_pyarrow"] = bool(pq.read_table(os.path.join(tmp, "enc.parquet")).equals(pq.read_table(os.path.join(tmp, "enc_pa.parquet"))))
        out["encode_one_table_of_n_rows"] = enc
        out["host_threads"] = os.cpu_count()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
