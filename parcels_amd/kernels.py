"""Built-in kernels (names and signatures of src/parcels/kernels/_advection.py and _advectiondiffusion.py).

These functions are *tokens*: ``Kernel`` recognises them by identity (the reference does the same for
AdvectionRK45, kernel.py:129-134) and maps each to the PK_KERNEL_* id of the HIP implementation in
csrc/pk_kernels.h.  Their bodies never run; calling one directly is an error, because this package has no
NumPy execution path.
"""

from __future__ import annotations

__all__ = [
    "AdvectionDiffusionEM",
    "AdvectionDiffusionM1",
    "AdvectionEE",
    "AdvectionRK2",
    "AdvectionRK2_3D",
    "AdvectionRK4",
    "AdvectionRK4_3D",
    "AdvectionRK45",
    "DeleteOutOfBounds",
    "DeleteParticle",
    "DiffusionUniformKh",
    "DoNothing",
    "MoveEast",
    "MoveNorth",
    "SampleField",
    "SubmergeParticle",
]


def _device_only(name):
    raise RuntimeError(
        f"{name} is a device kernel of parcels_amd: pass it to ParticleSet.execute(); it cannot be called on the host"
    )


def AdvectionEE(particles, fieldset):  # _advection.py:78-82
    """Explicit Euler advection."""
    _device_only("AdvectionEE")


def AdvectionRK2(particles, fieldset):  # _advection.py:21-28
    """Second-order Runge-Kutta advection."""
    _device_only("AdvectionRK2")


def AdvectionRK2_3D(particles, fieldset):  # _advection.py:31-39
    """Second-order Runge-Kutta advection including vertical velocity."""
    _device_only("AdvectionRK2_3D")


def AdvectionRK4(particles, fieldset):  # _advection.py:42-55
    """Fourth-order Runge-Kutta advection."""
    _device_only("AdvectionRK4")


def AdvectionRK4_3D(particles, fieldset):  # _advection.py:58-75
    """Fourth-order Runge-Kutta advection including vertical velocity."""
    _device_only("AdvectionRK4_3D")


def AdvectionRK45(particles, fieldset):  # _advection.py:85-155
    """Adaptive Runge-Kutta-Fehlberg 4(5) advection (needs RK45_tol/RK45_min_dt/RK45_max_dt and a next_dt Variable)."""
    _device_only("AdvectionRK45")


def AdvectionDiffusionM1(particles, fieldset):  # _advectiondiffusion.py:21-67
    """2-D advection-diffusion, Milstein scheme of order 1 (needs Kh_zonal, Kh_meridional, fieldset.dres)."""
    _device_only("AdvectionDiffusionM1")


def AdvectionDiffusionEM(particles, fieldset):  # _advectiondiffusion.py:70-117
    """2-D advection-diffusion, Euler-Maruyama scheme."""
    _device_only("AdvectionDiffusionEM")


def DiffusionUniformKh(particles, fieldset):  # _advectiondiffusion.py:120-153
    """2-D diffusion with uniform Kh (no advection)."""
    _device_only("DiffusionUniformKh")


# Native forms of the recovery kernels that the reference's tests write in Python and append to the kernel list.
def DeleteParticle(particles, fieldset):  # tests/common_kernels.py:12-13
    """state >= 50 -> Delete."""
    _device_only("DeleteParticle")


def DeleteOutOfBounds(particles, fieldset):  # tests/test_advection.py:157-161
    """ErrorOutOfBounds / ErrorThroughSurface -> Delete."""
    _device_only("DeleteOutOfBounds")


def DoNothing(particles, fieldset):  # tests/common_kernels.py:8-9
    """Time passes, nothing moves (the kernel of the reference's loop and output tests)."""
    _device_only("DoNothing")


def MoveEast(particles, fieldset):  # tests/common_kernels.py:16-17
    """particles.dx += 0.1"""
    _device_only("MoveEast")


def MoveNorth(particles, fieldset):  # tests/common_kernels.py:20-21
    """particles.dy += 0.1"""
    _device_only("MoveNorth")


def SubmergeParticle(particles, fieldset):  # tests/test_advection.py:163-174
    """ErrorThroughSurface -> resample UV, dz = 0, z = 0, state = Evaluate."""
    _device_only("SubmergeParticle")


def SampleField(field: str, into):
    """The user kernel every Parcels tutorial writes,

        def SampleP(particles, fieldset):
            particles.p = fieldset.P[particles]

    as a device kernel: ``pset.execute([AdvectionRK4, SampleField("P", into="p")], ...)`` samples scalar field ``field`` at every
    particle's (t, z, y, x) in each step of the kernel loop (kernel.py:206-216) and stores it in the particle Variable ``into``
    (float32 or float64, added with ``Particle.add_variable``), with the reference's status-code side effects of a failed
    sample (field.py:307-378).  The vector form ``SampleField("UV", into=("u", "v"))`` / ``SampleField("UVW", into=("u", "v", "w"))``
    is ``particles.u, particles.v = fieldset.UV[particles]`` (tests/test_particleset_execute.py:195-243: the converted velocity
    components of VectorField.__getitem__, field.py:250-304); ``None`` in the tuple discards a component like ``_`` does.
    Returns a kernel token named ``Sample<field>``."""
    if isinstance(into, (tuple, list)):
        into = tuple(into)
        if not all(v is None or isinstance(v, str) for v in into) or all(v is None for v in into):
            raise TypeError("SampleField(vector_field_name, into=(variable_name | None, ...))")
    if not (isinstance(field, str) and isinstance(into, (str, tuple))):
        raise TypeError("SampleField(field_name, into=variable_name)")

    def token(particles, fieldset):
        _device_only(token.__name__)

    token.__name__ = token.__qualname__ = f"Sample{field}"
    token.__doc__ = f"particles.{into} = fieldset.{field}[particles]"
    token._pk_sample = (field, into)
    return token


def kernel_id(f):
    """PK_KERNEL_* id of a kernel token, or None for a function this package cannot run."""
    if getattr(f, "_pk_sample", None) is not None:
        return 10  # PK_KERNEL_SAMPLE_FIELD
    return KERNEL_IDS.get(f)


KERNEL_IDS = {
    AdvectionEE: 1,
    AdvectionRK2: 2,
    AdvectionRK2_3D: 3,
    AdvectionRK4: 4,
    AdvectionRK4_3D: 5,
    AdvectionRK45: 6,
    AdvectionDiffusionM1: 7,
    AdvectionDiffusionEM: 8,
    DiffusionUniformKh: 9,
    DeleteParticle: 20,
    DeleteOutOfBounds: 21,
    SubmergeParticle: 22,
    DoNothing: 23,
    MoveEast: 24,
    MoveNorth: 25,
}
