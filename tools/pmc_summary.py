#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of bench.py into
profiles/<name>.md + profiles/pmc_latest.json.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE/WRITE_SIZE are in KiB of 64-B fabric requests;
FETCH_SIZE reports exactly half of the bytes of a wide coalesced read on this rocprofv3, other access widths are
uncalibrated.  We therefore calibrate on the float4 copy kernel that bench.py runs (pk::copy_kernel: exactly 1 GiB read and
1 GiB written per dispatch) and scale the advection kernel's counters by the same factors.
"""
import csv
import glob
import json
import sys


def read_counter(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return rows


def main(fetch_dir, write_dir, out_md, particles, steps):
    res = {}
    for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        rows = read_counter(d, name)
        adv = [v for k, v in rows if "advect_kernel" in k]
        cpy = [v for k, v in rows if "copy_kernel" in k]
        res[name] = {"advect": adv, "copy": cpy}
    GiB = float(1 << 30)
    # calibration factors: true bytes / (counter * 1024)
    f_cal = GiB / (1024.0 * (sum(res["FETCH_SIZE"]["copy"]) / max(len(res["FETCH_SIZE"]["copy"]), 1))) if res["FETCH_SIZE"]["copy"] else None
    w_cal = GiB / (1024.0 * (sum(res["WRITE_SIZE"]["copy"]) / max(len(res["WRITE_SIZE"]["copy"]), 1))) if res["WRITE_SIZE"]["copy"] else None
    fetch_adv = res["FETCH_SIZE"]["advect"][-1] if res["FETCH_SIZE"]["advect"] else None  # last dispatch = the timed launch
    write_adv = res["WRITE_SIZE"]["advect"][-1] if res["WRITE_SIZE"]["advect"] else None
    lines = ["# HBM traffic of the advection kernel from rocprofv3 PMC passes", "",
             f"copy_kernel (1 GiB in, 1 GiB out per dispatch): FETCH_SIZE = {res['FETCH_SIZE']['copy']}, WRITE_SIZE = {res['WRITE_SIZE']['copy']} (KiB)",
             f"calibration factors true/(counter*1024): fetch {f_cal}, write {w_cal}", "",
             f"advect_kernel dispatches: FETCH_SIZE = {res['FETCH_SIZE']['advect']}, WRITE_SIZE = {res['WRITE_SIZE']['advect']} (KiB)"]
    out = {"particles_per_gpu": particles, "steps": steps, "source": out_md}
    if fetch_adv is not None and write_adv is not None:
        fb = fetch_adv * 1024.0 * (f_cal or 1.0)
        wb = write_adv * 1024.0 * (w_cal or 1.0)
        out["traffic_bytes_per_launch"] = fb + wb
        out["fetch_bytes"] = fb
        out["write_bytes"] = wb
        lines += ["", f"timed launch ({particles} particles x {steps} steps): fetch {fb/1e9:.3f} GB + write {wb/1e9:.3f} GB = {(fb+wb)/1e9:.3f} GB "
                  f"= {(fb+wb)/(particles*steps):.1f} B per particle-step (algorithmic: 1112 B)"]
    open(out_md, "w").write("\n".join(lines) + "\n")
    json.dump(out, open("profiles/pmc_latest.json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(float(sys.argv[4])), int(sys.argv[5]))
