#!/usr/bin/env python
"""Summarise the rocprofv3 passes that tools/gpu_profile_c2.sh leaves under gpurun_out/ into profiles/<name>.md and
profiles/pmc_latest.json (which bench.py folds into its `roofline` object).

    python tools/pmc_summary.py TAG NAME [--particles 1e7 --steps 24]
        reads  gpurun_out/TAG_pmc_{sq,sq2,fetch,write}/p_counter_collection.csv, gpurun_out/TAG_trace/, gpurun_out/TAG_bench.json
        writes profiles/NAME_pmc.md, profiles/NAME_kernel_trace.md, profiles/NAME_bench.json, profiles/pmc_latest.json

Counter conventions on gfx950 (MI355X_MICROARCH.md): the passes are separate runs with --kernel-trace only; SQ_ACTIVE_INST_* and
SQ_WAVE_CYCLES / SQ_WAIT_* count QUAD-cycles (x4 = cycles); GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE / WRITE_SIZE
are KiB of 64-byte fabric requests and FETCH_SIZE reports half the bytes of a wide coalesced read -- both are calibrated here on
the float4 copy kernel bench.py runs first (pk::copy_kernel: exactly 1 GiB read + 1 GiB written per dispatch).
Everything is reported PER PARTICLE-STEP as well, so that bench.py can scale it to any --steps / --particles.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD, N_XCD = 1024, 8


def source_hash():
    """sha256 over the kernel sources: bench.py marks the counters stale when the kernels changed since they were collected."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "parcels_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "parcels_amd", "csrc", "*.hip"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def code_hash(kernel_name):
    """machine-code hash of the profiled kernel in the library that was profiled (tools/kernel_code_hash.py, written by `make`)"""
    try:
        for v in json.load(open(os.path.join(ROOT, "parcels_amd", "kernel_code_hashes.json"))).values():
            if v.get("kernel") == kernel_name:
                return v.get("code_hash")
    except Exception:
        pass
    return None


MATCH = "advect"  # --match: substring that selects the advection kernel of interest (a run of config c5 launches two programs)


def read_pass(d):
    """-> {kernel class: [ {counter: value, ...} per dispatch ]} for the advection and the copy kernel"""
    out = {"advect": {}, "copy": {}}
    meta = {}
    f = os.path.join(d, "p_counter_collection.csv")
    if not os.path.exists(f):
        return out, meta
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        cls = "advect" if ("advect" in k and MATCH in k) else ("copy" if "copy_kernel" in k else None)
        if cls is None:
            continue
        disp = int(r["Dispatch_Id"])
        out[cls].setdefault(disp, {})
        out[cls][disp][r["Counter_Name"]] = out[cls][disp].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if cls == "advect":
            meta = {"kernel": k, "vgpr": int(r["VGPR_Count"]), "accum_vgpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                    "lds": int(r["LDS_Block_Size"]), "scratch": int(r["Scratch_Size"])}
    return out, meta


def trace_summary(tag, out_md):
    dbs = glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_trace", "**", "*.db"), recursive=True)
    if not dbs:
        return None
    cur = sqlite3.connect(dbs[0]).cursor()
    lines = [f"# rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline` ({tag})", "", "## top kernels (us)", "",
             "| kernel | calls | total_us | avg_us | % |", "|---|---|---|---|---|"]
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name if len(name) < 110 else name[:70] + " ... " + name[-30:]
        lines.append(f"| `{short}` | {calls} | {total:.1f} | {avg:.1f} | {pct:.2f} |")  # the stats view is in microseconds already
    lines += ["", "## dispatches of the advection / sort kernels (the LAST advect dispatch is the timed launch)", "",
              "| kernel | duration_us | grid | wg | lds | scratch | vgpr | agpr | sgpr |", "|---|---|---|---|---|---|---|---|---|"]
    q = ("select name,duration,grid_x,workgroup_x,lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count from kernels "
         "where name like '%pk::advect%' or name like '%pk::sort_key%' order by start")
    last = None
    for r in cur.execute(q):
        lines.append(f"| `{r[0][:90]}` | {r[1] / 1e3:.1f} | {r[2]} | {r[3]} | {r[4]} | {r[5]} | {r[6]} | {r[7]} | {r[8]} |")
        if "advect" in r[0]:
            last = r[1] / 1e6
    open(out_md, "w").write("\n".join(lines) + "\n")
    return last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("name")
    ap.add_argument("--particles", type=float, default=1e7)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--evals-per-step", type=int, default=4)
    ap.add_argument("--no-latest", action="store_true", help="do not rewrite profiles/pmc_latest.json (passes of tools/gpu_profile_cfg.sh: C3 / C5, not the bench workload)")
    ap.add_argument("--psteps", type=float, default=0, help="particle-steps of the profiled launch when it is not particles x steps (deleted particles)")
    ap.add_argument("--match", default="advect", help="substring of the kernel name to summarise (e.g. rk45_kernel, m1_kernel)")
    ap.add_argument("--secondary", default=None, help="also record this kernel under that key in profiles/pmc_secondary_latest.json (bench.py's `secondary` reads it)")
    a = ap.parse_args()
    global MATCH
    MATCH = a.match
    g = os.path.join(ROOT, "gpurun_out", a.tag + "_")
    npart, K = int(a.particles), a.steps
    psteps = int(a.psteps) if a.psteps else npart * K
    wave_evals = psteps * a.evals_per_step / 64
    c = {}
    meta = {}
    copy = {}
    for p in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
        d, m = read_pass(g + p)
        if d["advect"]:
            c.update(d["advect"][max(d["advect"])])  # last dispatch = the timed launch
            meta = m or meta
        for disp in d["copy"].values():
            for k, v in disp.items():
                copy.setdefault(k, []).append(v)
    GiB = float(1 << 30)
    f_cal = GiB / (1024.0 * (sum(copy["FETCH_SIZE"]) / len(copy["FETCH_SIZE"]))) if copy.get("FETCH_SIZE") else None
    w_cal = GiB / (1024.0 * (sum(copy["WRITE_SIZE"]) / len(copy["WRITE_SIZE"]))) if copy.get("WRITE_SIZE") else None
    fetch_b = c["FETCH_SIZE"] * 1024.0 * (f_cal or 2.0) if "FETCH_SIZE" in c else None
    write_b = c["WRITE_SIZE"] * 1024.0 * (w_cal or 1.0) if "WRITE_SIZE" in c else None
    trace_ms = trace_summary(a.tag, os.path.join(ROOT, "profiles", a.name + "_kernel_trace.md"))
    cyc_xcd = c.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD
    valu_busy_cyc = c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0
    out = {
        "source": f"profiles/{a.name}_pmc.md", "source_hash": source_hash(), "code_hash": code_hash(meta.get("kernel")), "particles_per_gpu": npart, "steps": K, **meta,
        "per_particle_step": {
            "valu_busy_simd_cycles": valu_busy_cyc / psteps,
            "valu_insts_per_wave_eval": c.get("SQ_INSTS_VALU", 0.0) / wave_evals,
            "salu_insts_per_wave_eval": c.get("SQ_INSTS_SALU", 0.0) / wave_evals,
            "fetch_bytes": fetch_b / psteps if fetch_b is not None else None,
            "write_bytes": write_b / psteps if write_b is not None else None,
        },
        "profiled_launch": {"gpu_cycles_per_xcd": cyc_xcd, "valu_utilisation_at_measured_clock": valu_busy_cyc / (N_SIMD * cyc_xcd) if cyc_xcd else None,
                            "kernel_ms_in_trace_pass": trace_ms * 1e3 if trace_ms else None},
        "counters": c,
    }
    if fetch_b is not None and write_b is not None:
        out["traffic_bytes_per_launch"] = fetch_b + write_b
    if not a.no_latest:
        json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
    if a.secondary:
        sp = os.path.join(ROOT, "profiles", "pmc_secondary_latest.json")
        sec = json.load(open(sp)) if os.path.exists(sp) else {}
        pps = out["per_particle_step"]
        sec[a.secondary] = {"source": out["source"], "source_hash": out["source_hash"], "code_hash": out.get("code_hash"), "kernel": meta.get("kernel"), "scratch": meta.get("scratch"),
                            "hbm_bytes_per_particle_step": (pps["fetch_bytes"] + pps["write_bytes"]) if pps["fetch_bytes"] is not None and pps["write_bytes"] is not None else None,
                            "valu_insts_per_wave_eval": pps["valu_insts_per_wave_eval"], "salu_insts_per_wave_eval": pps["salu_insts_per_wave_eval"],
                            "valu_utilisation_at_measured_clock": out["profiled_launch"]["valu_utilisation_at_measured_clock"]}
        json.dump(sec, open(sp, "w"), indent=1)
    L = [f"# PMC counters of the timed advection launch ({a.tag}): {npart} particles x {K} steps", "",
         f"kernel `{meta.get('kernel', '?')[:120]}`: arch VGPR {meta.get('vgpr')} (+{meta.get('accum_vgpr')} acc), SGPR {meta.get('sgpr')}, LDS {meta.get('lds')} B, scratch {meta.get('scratch')} B/lane", "",
         "| counter | value | per wave-evaluation |", "|---|---|---|"]
    for k in sorted(c):
        L.append(f"| {k} | {c[k]:.4g} | {c[k] / wave_evals:.2f} |")
    L += ["", f"* VALU instructions per wave-evaluation: **{c.get('SQ_INSTS_VALU', 0) / wave_evals:.0f}** (SALU {c.get('SQ_INSTS_SALU', 0) / wave_evals:.0f}); "
              f"VALU busy = SQ_ACTIVE_INST_VALU x 4 = {valu_busy_cyc:.4g} SIMD-cycles = {valu_busy_cyc / max(c.get('SQ_INSTS_VALU', 1), 1):.2f} cycles per instruction",
          f"* GPU cycles of the launch (GRBM_GUI_ACTIVE / 8 XCDs): {cyc_xcd:.4g} -> VALU utilisation {valu_busy_cyc / (N_SIMD * cyc_xcd) if cyc_xcd else float('nan'):.3f} of {N_SIMD} SIMDs at the clock the profiled pass ran at",
          f"* wave-cycle split: waiting on memory / LDS (SQ_WAIT_ANY) {c.get('SQ_WAIT_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f}, issue stall (SQ_WAIT_INST_ANY) "
          f"{c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f}, issuing (SQ_ACTIVE_INST_ANY) {c.get('SQ_ACTIVE_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):.2f}"]
    if fetch_b is not None and write_b is not None:
        cal = (f"copy_kernel calibration (1 GiB in + 1 GiB out per dispatch): FETCH_SIZE {copy.get('FETCH_SIZE')} KiB -> x{f_cal:.4f}, WRITE_SIZE {copy.get('WRITE_SIZE')} KiB -> x{w_cal:.4f}"
               if f_cal and w_cal else "no copy kernel in this run: FETCH_SIZE x2, WRITE_SIZE x1 as calibrated on the bench passes of the same binary (the *_c2_pmc.md of the same round letter)")
        L += [f"* {cal}",
              f"* HBM traffic of the launch: fetch {fetch_b / 1e9:.3f} GB + write {write_b / 1e9:.3f} GB = {(fetch_b + write_b) / 1e9:.3f} GB = "
              f"**{(fetch_b + write_b) / psteps:.1f} B per particle-step** (algorithmic model of SURVEY 8d for C2: 1112 B, served by L2 / Infinity Cache)"]
    open(os.path.join(ROOT, "profiles", a.name + "_pmc.md"), "w").write("\n".join(L) + "\n")
    for suffix in ("bench.json", "run.json"):
        if os.path.exists(g + suffix):
            open(os.path.join(ROOT, "profiles", a.name + "_" + suffix), "w").write(open(g + suffix).read())
    print("\n".join(L))


if __name__ == "__main__":
    main()
