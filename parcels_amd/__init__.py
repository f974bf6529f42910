"""parcels_amd -- MI355X-native engine behind the Parcels ``ParticleSet.execute()`` hot path.

Python keeps the reference's API surface (FieldSet / ParticleSet / Kernel / StatusCode / built-in kernels); the
time loop, cell search and interpolation run in hand-written HIP kernels (csrc/) behind the C ABI of
include/parcels_hip.h.  There is no NumPy/CPU execution path.
"""

from . import convert, kernels
from .dataset import DataArray, Dataset
from .field import Field, FieldEvalWarning, TimeInterval, VectorField
from .fieldset import FieldSet
from .interpolators import (
    CGrid_Tracer,
    CGrid_Velocity,
    XConstantField,
    XFreeslip,
    XLinear,
    XLinear_Velocity,
    XLinearInvdistLandTracer,
    XNearest,
    XPartialslip,
)
from .kernel import Kernel, KernelWarning
from .kernels import (
    AdvectionDiffusionEM,
    AdvectionDiffusionM1,
    AdvectionEE,
    AdvectionRK2,
    AdvectionRK2_3D,
    AdvectionRK4,
    AdvectionRK4_3D,
    AdvectionRK45,
    DeleteOutOfBounds,
    DeleteParticle,
    DiffusionUniformKh,
    DoNothing,
    MoveEast,
    MoveNorth,
    SampleField,
    SubmergeParticle,
)
from .particle import Particle, ParticleClass, Variable, get_default_particle
from .particlefile import ParticleFile, read_particlefile
from .compat_v3 import particlefile_to_v3_zarr
from .particleset import ParticleSet, ParticleSetWarning
from .sgrid import FaceNodePadding, Padding, SGrid2DMetadata
from .sources import LevelSource, NetCDFLevels, NpyLevels, ZarrLevels, read_netcdf_variable
from .statuscodes import (
    AllParcelsErrorCodes,
    FieldInterpolationError,
    FieldOutOfBoundError,
    FieldOutOfBoundSurfaceError,
    FieldSamplingError,
    GeneralError,
    GridSearchingError,
    KernelError,
    OutsideTimeInterval,
    StatusCode,
)
from .xgrid import SphericalMesh, XGrid

__version__ = "0.1.0"
