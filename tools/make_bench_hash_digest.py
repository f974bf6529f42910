#!/usr/bin/env python
"""Digest of the spatial-hash table the HOST builder (parcels_amd/spatialhash.py: the restatement pinned to the reference's table by
tests/test_spatialhash_reference.py) produces for a bench mesh -> tools/bench_hash_digests.json.  tools/bench_configs.py compares the
device-built table of a full-size run with it before the oracle may use that table (independent_hash_table).  CPU only; the
4322 x 3059 mesh takes a few minutes and ~20 GB of host memory.

    python tools/make_bench_hash_digest.py [--scale 1.0]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    a = ap.parse_args()
    from parcels_amd import spatialhash as sh
    from tools.bench_configs import HASH_DIGESTS, nemo_like_coords

    nx, ny = max(int(4322 * a.scale), 32), max(int(3059 * a.scale), 32)
    lon, lat, _, _ = nemo_like_coords(nx, ny)
    t0 = time.perf_counter()
    h = sh.SpatialHash(lon, lat, True)
    el = time.perf_counter() - t0
    d = json.load(open(HASH_DIGESTS)) if os.path.exists(HASH_DIGESTS) else {}
    d[f"{nx}x{ny}"] = {"checksum": h.checksum(), "builder": "parcels_amd.spatialhash.SpatialHash (host, NumPy)", "build_s": round(el, 1),
                      "mesh": "tools/bench_configs.py: nemo_like_coords"}
    json.dump(d, open(HASH_DIGESTS, "w"), indent=1, sort_keys=True)
    print(json.dumps(d[f"{nx}x{ny}"]))


if __name__ == "__main__":
    main()
