#!/usr/bin/env python
"""Write-out off the critical path: the C2 workload of bench.py with a ParticleFile written every `--every` steps, once with the
inline path (D2H of the to-write columns + write filter + Parquet encode between two launches) and once with the double-buffered
asynchronous path (device snapshot -> D2H on the copy stream -> writer thread), same file byte for byte.  Prints one JSON object."""
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--every", type=int, default=2)
    a = ap.parse_args()
    import torch

    import parcels_amd as pa
    from bench import c2_case
    from tests.case_utils import build_fieldset, build_pset

    n = int(a.particles)
    case = c2_case(seed=1, lo=0, hi=n)
    fs = build_fieldset(case)
    fs.to_device()
    out = {"workload": f"C2, {n} fp64 particles, AdvectionRK4, {a.steps} steps, ParticleFile every {a.every} steps ({a.steps // a.every + 1} tables of {n} rows)"}
    files = {}
    with tempfile.TemporaryDirectory() as tmp:
        for mode in ("warmup", "inline", "async"):
            pset = build_pset(case, fs, sort_by_cell=True)
            pset.async_output = mode == "async"
            path = os.path.join(tmp, f"{mode}.parquet")
            pf = pa.ParticleFile(path, outputdt=float(a.every * case["dt"]), compression="zstd")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pset.execute(pa.AdvectionRK4, dt=case["dt"], runtime=a.steps * case["dt"], output_file=pf)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            if mode != "warmup":
                files[mode] = open(path, "rb").read()
                out[mode] = {"wall_s": wall, "file_MB": len(files[mode]) / 1e6}
        pset = build_pset(case, fs, sort_by_cell=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pset.execute(pa.AdvectionRK4, dt=case["dt"], runtime=a.steps * case["dt"])
        torch.cuda.synchronize()
        out["no_output"] = {"wall_s": time.perf_counter() - t0}
        # the Parquet encode of ONE table of n rows (the default Variables: particle_id, t, z, y, x), multi-threaded writer vs pyarrow's
        import pyarrow as pyarrow
        import pyarrow.parquet as pq

        from parcels_amd.parquet_writer import FastParquetWriter

        cols = {k: np.ascontiguousarray(pset._data[k]) for k in ("particle_id", "t", "z", "y", "x")}
        schema = pyarrow.schema([pyarrow.field(k, pyarrow.from_numpy_dtype(v.dtype)) for k, v in cols.items()])
        enc = {}
        for label, threads in (("fast_writer_1_thread", 1), ("fast_writer_8_threads", 8), ("fast_writer_32_threads", 32)):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                with FastParquetWriter(os.path.join(tmp, "enc.parquet"), schema, compression="zstd", threads=threads) as w:
                    w.write_columns(cols)
                best = min(best, time.perf_counter() - t0)
            enc[label] = best
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            with pq.ParquetWriter(os.path.join(tmp, "enc_pa.parquet"), schema, compression="zstd", use_dictionary=False) as w:
                w.write_table(pyarrow.table(cols, schema=schema))
            best = min(best, time.perf_counter() - t0)
        enc["pyarrow_writer"] = best
        enc["same_table"] = bool(pq.read_table(os.path.join(tmp, "enc.parquet")).equals(pq.read_table(os.path.join(tmp, "enc_pa.parquet"))))
        out["encode_seconds_per_table_of_n_rows"] = enc
    out["byte_identical"] = files["inline"] == files["async"]
    out["write_out_cost_inline_s"] = out["inline"]["wall_s"] - out["no_output"]["wall_s"]
    out["write_out_cost_async_s"] = out["async"]["wall_s"] - out["no_output"]["wall_s"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
