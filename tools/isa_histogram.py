#!/usr/bin/env python
"""Instruction-class histogram of one kernel's inner loops from the compiler's own assembly (hipcc -S, gfx950), for an honest
VALU roofline: on CDNA4 a wave64 fp64 VALU instruction occupies its SIMD for 4 cycles (16 lanes per cycle: 78.6 TFLOP/s FMA peak =
1024 SIMDs x 2.4 GHz x 16 lanes x 2), every other VALU instruction for 2 (SIMD-32; MI355X_MICROARCH.md "v_fma_f32 (wave64) 2 cyc").

    python tools/isa_histogram.py pk_prog_rk4_fast.hip 'advect_fast_kernel<double, 0, false>' --name r03_c2 [--min-depth 2]

writes profiles/<name>_isa.md and profiles/<name>_isa.json (and profiles/isa_latest.json unless --no-latest), which bench.py folds
into its `roofline` object together with the dynamic instruction count of the rocprofv3 PMC pass (SQ_INSTS_VALU).

The histogram is STATIC: instructions of the basic blocks at loop depth >= --min-depth (the stage loop of the fused step loop and
everything inside it), each counted once.  The specialised variants of one evaluation (second time / depth level taking part or not)
are alternative straight-line blocks of the same mix, so the class FRACTIONS describe the executed stream well; the absolute count per
evaluation comes from the hardware counters, not from here.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "parcels_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-Wno-unused-function", "--offload-device-only", "-S"]

FP64_FMA = ("v_fma_f64", "v_fmac_f64")
FP64_ARITH = ("v_add_f64", "v_mul_f64", "v_min_f64", "v_max_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_scale_f64", "v_div_fmas_f64",
              "v_div_fixup_f64", "v_trig_preop_f64", "v_ldexp_f64", "v_frexp", "v_fract_f64", "v_rndne_f64", "v_floor_f64", "v_ceil_f64", "v_trunc_f64")


def classify(op):
    if not op.startswith("v_"):
        if op.startswith("s_"):
            if op.startswith(("s_load", "s_buffer_load")):
                return "smem"
            if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio", "s_sched")):
                return "wait/nop"
            if "branch" in op or op.startswith("s_endpgm"):
                return "branch"
            return "salu"
        if op.startswith("ds_"):
            return "lds"
        if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            return "vmem"
        return "other"
    if op.startswith(FP64_FMA):
        return "valu fp64 fma"
    if op.startswith(FP64_ARITH):
        return "valu fp64 arith"
    if op.startswith("v_cmp") and "f64" in op:
        return "valu fp64 compare"
    if op.startswith("v_cvt"):
        return "valu convert"
    if op.startswith(("v_mov", "v_cndmask", "v_readlane", "v_writelane", "v_readfirstlane", "v_accvgpr", "v_swap", "v_perm", "v_bfi", "v_mbcnt")):
        return "valu move/select"
    if "f32" in op or "f16" in op:
        return "valu fp32"
    return "valu integer/address"


CYCLES = {"valu fp64 fma": 4, "valu fp64 arith": 4, "valu fp64 compare": 4, "valu convert": 4,  # f64 <-> f32/i32 conversions run at the fp64 rate
          "valu move/select": 2, "valu fp32": 2, "valu integer/address": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tu")
    ap.add_argument("kernel", help="demangled name fragment, e.g. 'advect_fast_kernel<double, 0, false>'")
    ap.add_argument("--name", required=True)
    ap.add_argument("--min-depth", type=int, default=2)
    ap.add_argument("--no-latest", action="store_true")
    ap.add_argument("flags", nargs="*")
    a = ap.parse_args()
    asm = f"/tmp/isa_{os.getpid()}.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, *a.flags, "-c", a.tu, "-o", asm], cwd=CSRC, stderr=subprocess.DEVNULL)
    txt = open(asm).read()
    os.remove(asm)
    syms = re.findall(r"^(_Z\w+):", txt, re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.splitlines()
    match = [s for s, d in zip(syms, dem) if a.kernel in d and ".kd" not in s]
    if len(match) != 1:
        sys.exit(f"kernel fragment matches {len(match)} symbols: {[d for d in dem if a.kernel.split('<')[0] in d][:8]}")
    sym = match[0]
    body = txt[txt.index(sym + ":"):]
    body = body[: body.index(".Lfunc_end")]
    depth = 0
    hist = collections.Counter()
    ops = collections.Counter()
    all_hist = collections.Counter()
    for line in body.splitlines():
        m = re.search(r"Depth=(\d+)", line)
        if re.match(r"^(\.LBB\w+:|; %bb\.\d+:)", line):
            depth = int(m.group(1)) if m else 0
            continue
        if m and line.lstrip().startswith(";"):  # "=> This Inner Loop Header: Depth=k" on a continuation comment line
            depth = max(depth, int(m.group(1)))
            continue
        if not line.startswith("\t") or line.strip().startswith((".", ";")):
            continue
        op = line.split()[0]
        c = classify(op)
        all_hist[c] += 1
        if depth >= a.min_depth:
            hist[c] += 1
            ops[op] += 1
    valu = {k: v for k, v in hist.items() if k.startswith("valu")}
    nv = sum(valu.values())
    frac = {k: v / nv for k, v in valu.items()}
    cyc_per_inst = sum(frac[k] * CYCLES[k] for k in frac)
    fp64_arith_frac = frac.get("valu fp64 fma", 0) + frac.get("valu fp64 arith", 0)
    flops_per_valu_inst = 2 * frac.get("valu fp64 fma", 0) + frac.get("valu fp64 arith", 0)
    out = {"kernel": dem[syms.index(sym)], "tu": a.tu, "min_depth": a.min_depth, "static_instructions_in_loops": dict(hist), "static_instructions_whole_kernel": dict(all_hist),
           "valu_class_fractions": frac, "valu_cycles_per_instruction": cyc_per_inst, "fp64_arith_fraction_of_valu": fp64_arith_frac,
           "fp64_flops_per_valu_instruction_per_lane": flops_per_valu_inst, "cycles_model": CYCLES,
           "top_ops": ops.most_common(25)}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", a.name + "_isa.json"), "w"), indent=1)
    if not a.no_latest:
        json.dump(out, open(os.path.join(ROOT, "profiles", "isa_latest.json"), "w"), indent=1)
    L = [f"# ISA histogram of `{out['kernel']}` ({a.tu}), basic blocks at loop depth >= {a.min_depth}", "",
         "| class | static instructions | share of VALU | SIMD cycles each (wave64) |", "|---|---|---|---|"]
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
        L.append(f"| {k} | {v} | {frac[k]:.3f} | {CYCLES[k]} |" if k in frac else f"| {k} | {v} | | |")
    L += ["", f"* VALU instructions: {nv}; cycle-weighted cost {cyc_per_inst:.2f} SIMD-cycles per VALU instruction (a flat 4 would overcharge by {4 / cyc_per_inst:.2f}x)",
          f"* fp64 arithmetic (add / mul / fma / division and square-root sequences): {fp64_arith_frac:.3f} of the VALU instructions, "
          f"{flops_per_valu_inst:.3f} flop per VALU instruction and lane (an FMA counts 2)",
          f"* conversions: {frac.get('valu convert', 0):.3f}; integer / address: {frac.get('valu integer/address', 0):.3f}; moves / selects: {frac.get('valu move/select', 0):.3f}",
          "", "most frequent opcodes: " + ", ".join(f"`{o}` {n}" for o, n in ops.most_common(25))]
    open(os.path.join(ROOT, "profiles", a.name + "_isa.md"), "w").write("\n".join(L) + "\n")
    print("\n".join(L))


if __name__ == "__main__":
    main()
