"""Interpolator markers (mirrors src/parcels/interpolators/_xinterpolators.py class names).

The arithmetic lives in the HIP kernels (csrc/pk_device.h: xlinear, cgrid_velocity); these classes select it,
exactly as assigning ``Field.interp_method`` does in the reference (field.py:130-135, 244-248).
"""


class ScalarInterpolator:
    kind = None

    def __repr__(self):  # interpolators/_base.py:10-11
        return f"{self.__class__.__name__}(...)"


class VectorInterpolator:
    kind = None

    def __repr__(self):  # interpolators/_base.py:19-20
        return f"{self.__class__.__name__}(...)"


class XLinear(ScalarInterpolator):  # _xinterpolators.py:112-153
    kind = 0


class XConstantField(ScalarInterpolator):  # _xinterpolators.py:156-166
    kind = 1


class XNearest(ScalarInterpolator):  # _xinterpolators.py:505-553
    kind = 2


class CGrid_Tracer(ScalarInterpolator):  # noqa: N801  _xinterpolators.py:335-383
    kind = 3


class XLinearInvdistLandTracer(ScalarInterpolator):  # _xinterpolators.py:556-613
    kind = 4


class XLinear_Velocity(VectorInterpolator):  # noqa: N801  _xinterpolators.py:169-190
    kind = 0


class CGrid_Velocity(VectorInterpolator):  # noqa: N801  _xinterpolators.py:193-332
    kind = 1


class XFreeslip(VectorInterpolator):  # _xinterpolators.py:480-490 (free-slip boundary condition, a = 1, b = 0)
    kind = 2


class XPartialslip(VectorInterpolator):  # _xinterpolators.py:493-502 (partial slip, a = b = 0.5)
    kind = 3
