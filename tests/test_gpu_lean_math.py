"""The routines the dedicated kernels use instead of the library's correctly rounded ones since round 6 (csrc/pk_fast_agrid.h: sqrt_lean,
div_lean, rcp_lean; csrc/pk_fast_cgrid.h: sincos_near), measured ON THE DEVICE against the library's over 1e8 random operands of the
magnitudes the evaluation feeds them (tools/lean_math_check.hip includes the shipped headers): square root and quotient within 1 ulp
(observed: 0 -- the same bits), a product with the shared reciprocal within 2 ulp (observed 1), sines / cosines near a known angle within
4 * 2^-53 absolute (observed 3, including the rounding of the angle sum the library routine is given)."""

from __future__ import annotations

import json
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lean_routines_on_the_device(gpu, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "lean_math_check")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function", "-I",
                           os.path.join(ROOT, "parcels_amd", "csrc"), os.path.join(ROOT, "tools", "lean_math_check.hip"), "-o", exe],
                          stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0, out
    assert out["samples"] >= 1e8
    assert out["sqrt_lean_max_ulp"] <= 1.0 and out["div_lean_max_ulp"] <= 1.0 and out["mul_rcp_lean_max_ulp"] <= 2.0, out
    assert out["sincos_near_max_abs_err_in_2^-53"] <= 4.0, out
