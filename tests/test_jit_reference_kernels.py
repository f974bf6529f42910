"""CPU: the translator over the user-written kernels of the reference's OWN tests and tutorials (tools/survey_user_kernels.py extracts every
`def f(particles, fieldset)` from /root/reference and gives it the Variables, fields and module constants it mentions).  A guard on
coverage -- the count may only go up -- and on honesty: every kernel that is left to the host path is left for a reason on the list."""
import os
import sys

import pytest

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")

KNOWN_REASONS = (
    "call of something other than a supported numpy function",  # gsw, neighbour search, fieldset.func(..), np.random, unravel_index
    "`if` on something other than a constant of the run",
    "sampling a velocity component by itself",
    "vector field '",                                            # VectorFields under other names than UV / UVW
    "statement Expr",                                            # pfile.write(..) from inside the kernel
    "statement FunctionDef",                                     # (kernel factories of the tests: the survey sees the outer function)
    "np.unique", "np.argwhere", "expression ListComp",           # Python loops over groups / rows
    "PARCELS_AMD_JIT_LIBM",                                      # transcendental functions: only on request
    "StatusCode.Success stored into particles.state",
    "outside the selection whose emptiness an `if` tests",       # `if np.any(mask): <something for ALL particles>`
    "arrays over different selections of the particles",         # (works in the reference's test because it has ONE particle)
)


def test_most_kernels_of_the_reference_compile():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import survey_user_kernels as sv

    rows = sv.survey()
    assert len(rows) >= 80
    errors = [r for r in rows if r[2] == "error"]
    assert not errors, errors
    translated = [r for r in rows if r[2] == "translated"]
    assert len(translated) >= 66, len(translated)
    for rel, name, status, reason in rows:
        if status != "translated":
            assert any(k in reason for k in KNOWN_REASONS), (rel, name, reason)
    names = {r[1] for r in translated}
    assert {"ArgoVerticalMovement", "KeepInOcean", "StopBelowBed", "AdvectionRK2_periodic", "DeleteParticle", "SampleT"} <= names
