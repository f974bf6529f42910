"""Particle status codes and the exceptions they raise (mirrors src/parcels/_core/statuscodes.py).

The integers are part of the device ABI (include/parcels_hip.h, PK_* state codes).
"""

__all__ = [
    "AllParcelsErrorCodes",
    "FieldInterpolationError",
    "FieldOutOfBoundError",
    "FieldOutOfBoundSurfaceError",
    "FieldSamplingError",
    "GeneralError",
    "GridSearchingError",
    "KernelError",
    "OutsideTimeInterval",
    "StatusCode",
]


class StatusCode:
    """statuscodes.py:19-34"""

    Success = 0
    EndofLoop = 1
    Evaluate = 10
    Repeat = 20
    Delete = 30
    StopExecution = 40
    StopAllExecution = 41
    Error = 50
    ErrorInterpolation = 51
    ErrorGridSearching = 52
    ErrorOutOfBounds = 60
    ErrorThroughSurface = 61
    ErrorOutsideTimeInterval = 70


class FieldInterpolationError(RuntimeError):
    """NaN field interpolation."""


class FieldOutOfBoundError(RuntimeError):
    """Field sampled out of bounds."""


class FieldOutOfBoundSurfaceError(RuntimeError):
    """Field sampled out of bounds at the surface."""


class FieldSamplingError(RuntimeError):
    """Field sampling error."""


class GridSearchingError(RuntimeError):
    """Grid searching error."""


class GeneralError(RuntimeError):
    """General error."""


class OutsideTimeInterval(RuntimeError):
    """Erroneous time extrapolation sampling."""

    def __init__(self, time, field=None):
        super().__init__(f"{field.name if field else 'Field'} sampled outside time domain at time {time}.")


class KernelError(RuntimeError):
    """General particles kernel error."""


AllParcelsErrorCodes = {
    FieldInterpolationError: StatusCode.ErrorInterpolation,
    FieldOutOfBoundError: StatusCode.ErrorOutOfBounds,
    FieldOutOfBoundSurfaceError: StatusCode.ErrorThroughSurface,
    GridSearchingError: StatusCode.ErrorGridSearching,
    OutsideTimeInterval: StatusCode.ErrorOutsideTimeInterval,
    KernelError: StatusCode.Error,
    GeneralError: StatusCode.Error,
}


def _raise_outside_time_interval_error(time, field=None):
    raise OutsideTimeInterval(time, field)


def _raise_field_out_of_bound_error(z, y, x):
    raise FieldOutOfBoundError(f"Field sampled out-of-bound, at (z={z}, y={y}, x={x})")


def _raise_field_out_of_bound_surface_error(z, y, x):
    raise FieldOutOfBoundSurfaceError(f"Field sampled out-of-bound at the surface, at (z={z}, y={y}, x={x})")


def _raise_field_interpolation_error(z, y, x):
    raise FieldInterpolationError(f"Field interpolation returned NaN at (z={z}, y={y}, x={x})")


def _raise_grid_searching_error(z, y, x):
    raise GridSearchingError(f"Grid searching failed at (z={z}, y={y}, x={x})")


def _raise_general_error(z, y, x):
    raise GeneralError(f"General error occurred at (z={z}, y={y}, x={x})")


# kernel.py:31-38 -- checked in this order after every Kernel.execute
ErrorsToThrow = {
    StatusCode.ErrorOutsideTimeInterval: _raise_outside_time_interval_error,
    StatusCode.ErrorOutOfBounds: _raise_field_out_of_bound_error,
    StatusCode.ErrorThroughSurface: _raise_field_out_of_bound_surface_error,
    StatusCode.ErrorInterpolation: _raise_field_interpolation_error,
    StatusCode.ErrorGridSearching: _raise_grid_searching_error,
    StatusCode.Error: _raise_general_error,
}
