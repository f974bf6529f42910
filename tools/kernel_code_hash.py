#!/usr/bin/env python
"""sha256 of the MACHINE CODE of the dedicated kernels, taken from the objects of the last `make` (parcels_amd/csrc/*.o).

    python tools/kernel_code_hash.py [--write]     ->  {kernel: hash}; --write: parcels_amd/kernel_code_hashes.json

The PMC summaries under profiles/ carry the hash of the kernel they profiled; bench.py marks counters stale when the kernel of the
library it runs has other code -- a change of the sources that leaves a kernel's instructions alone (another kernel's header, a
comment) does not.  Needs the ROCm LLVM tools (clang-offload-bundler, llvm-readelf); without them: {} (bench.py then falls back to
the hash of the sources)."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "parcels_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
# bench key -> (object, demangled kernel name)
KERNELS = {
    "AdvectionRK4": ("pk_prog_rk4_fast.o", "void pk::advect_fast_kernel<double, 0, false>(pk::KArgs)"),
    "AdvectionRK4_3D": ("pk_prog_cgrid_fast.o", "void pk::advect_cgrid_kernel<float, 0, true, true>(pk::KArgs)"),
    "AdvectionRK45": ("pk_prog_cgrid_fast.o", "void pk::advect_cgrid_rk45_kernel<float, 0, true>(pk::KArgs)"),
    "AdvectionDiffusionM1": ("pk_prog_cgrid_fast.o", "void pk::advect_cgrid_m1_kernel<float, 0, true>(pk::KArgs)"),
}


def code_object(obj, tmp):
    """the gfx950 code object of a host object: its .hip_fatbin section is an offload bundle"""
    bundle = os.path.join(tmp, os.path.basename(obj) + ".bundle")
    out = os.path.join(tmp, os.path.basename(obj) + ".co")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + bundle, obj, os.path.join(tmp, "unused.o")],
                   check=True, capture_output=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + bundle,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], check=True, capture_output=True)
    return out


def function_bytes(co):
    """{demangled name: bytes of the function} for every FUNC symbol of the code object"""
    sec = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-SW", co], check=True, capture_output=True, text=True).stdout
    text = None
    for l in sec.splitlines():
        m = re.match(r"\s*\[\s*\d+\]\s+\.text\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", l)
        if m:
            text = (int(m.group(1), 16), int(m.group(2), 16), int(m.group(3), 16))
    if text is None:
        return {}
    addr, off, size = text
    data = open(co, "rb").read()
    sym = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", "--demangle", co], check=True, capture_output=True, text=True).stdout
    out = {}
    for l in sym.splitlines():
        m = re.match(r"\s*\d+:\s+([0-9a-f]+)\s+(\d+)\s+FUNC\s+\S+\s+\S+\s+\S+\s+(.*)$", l)
        if m and int(m.group(2)) > 0:
            v, n = int(m.group(1), 16), int(m.group(2))
            out[m.group(3).strip()] = data[off + (v - addr): off + (v - addr) + n]
    return out


def hashes(csrc=CSRC):
    res = {}
    try:
        with tempfile.TemporaryDirectory() as tmp:
            cache = {}
            for key, (obj, name) in KERNELS.items():
                p = os.path.join(csrc, obj)
                if not os.path.exists(p):
                    continue
                if obj not in cache:
                    cache[obj] = function_bytes(code_object(p, tmp))
                b = cache[obj].get(name)
                if b:
                    res[key] = {"kernel": name, "code_bytes": len(b), "code_hash": hashlib.sha256(b).hexdigest()[:16]}
    except Exception as e:  # tools missing: no hashes
        print("kernel_code_hash:", e, file=sys.stderr)
        return {}
    return res


if __name__ == "__main__":
    csrc = CSRC
    for a in sys.argv[1:]:
        if a.startswith("--csrc="):
            csrc = a.split("=", 1)[1]
    h = hashes(csrc)
    if "--write" in sys.argv:
        json.dump(h, open(os.path.join(ROOT, "parcels_amd", "kernel_code_hashes.json"), "w"), indent=1)
    print(json.dumps(h, indent=1))
