"""Structured grid (mirrors the parts of src/parcels/_core/xgrid.py, mesh.py and basegrid.py the hot path reads).

Host-side metadata only: axes, ravel dimensions of ``ei``, C-grid offsets, coordinate arrays and the spatial-hash
table are handed to the device through ``pk_grid_create`` (include/parcels_hip.h).  Searching and interpolation run
in the HIP kernels.
"""

from __future__ import annotations

import numpy as np

from . import sgrid as _sgrid
from .dataset import Dataset
from .spatialhash import SpatialHash

EARTH_RADIUS = 6366707.019493707  # mesh.py:6


class FlatMesh:  # mesh.py:50-61
    radius = None

    def is_spherical(self):
        return False

    def __repr__(self):
        return "FlatMesh()"


class SphericalMesh:  # mesh.py:23-47
    def __init__(self, radius: float = EARTH_RADIUS):
        if not isinstance(radius, (int, float, np.number)):
            raise TypeError(f"radius must be a number, got {type(radius).__name__}")
        if radius <= 0:
            raise ValueError(f"radius must be positive, got {radius}")
        self.radius = radius

    @property
    def deg2m(self) -> float:
        return self.radius * np.pi / 180.0

    def is_spherical(self):
        return True

    def __repr__(self):
        return f"SphericalMesh(radius={self.radius})"


def get_mesh(mesh):  # mesh.py:66-73
    if isinstance(mesh, (SphericalMesh, FlatMesh)):
        return mesh
    if mesh == "flat":
        return FlatMesh()
    if mesh == "spherical":
        return SphericalMesh(EARTH_RADIUS)
    raise ValueError(f"mesh must be 'flat', 'spherical', or a SphericalMesh object. Got {mesh=!r}")


_XGRID_AXES_ORDERING = ("Z", "Y", "X")


class XGrid:
    """Rectilinear (1-D lon/lat) or curvilinear (2-D lon/lat) grid described by SGRID metadata."""

    def __init__(self, ds: Dataset, mesh="flat"):
        if ds.sgrid is None:
            raise ValueError("dataset carries no SGRID metadata (Dataset(sgrid=SGrid2DMetadata(...)))")
        self.sgrid_metadata: _sgrid.SGrid2DMetadata = ds.sgrid
        self._ds = ds
        self._mesh = get_mesh(mesh)
        self._spatialhash = None
        axes = self.axes
        if "X" in axes or "Y" in axes:
            lon, lat = self.lon, self.lat
            if lon.ndim != lat.ndim or lon.ndim not in (1, 2):
                raise ValueError("lon and lat must both be 1-D (rectilinear) or both 2-D (curvilinear)")
            if lon.ndim == 2 and lon.shape != lat.shape:
                raise ValueError("2-D lon and lat must have the same shape")
        if "Z" in axes:
            d = self.depth
            if d.ndim != 1:
                raise ValueError("depth must be 1-D")

    # -- axes and coordinates (xgrid.py:137-206) ----------------------------------------------------------------
    @property
    def axes(self):
        d2a = self.sgrid_metadata.dim_to_axis()
        present = {axis for dim, axis in d2a.items() if dim in self._ds.dims}
        return sorted(present, key=_XGRID_AXES_ORDERING.index)

    @property
    def lon(self):
        if "X" not in self.axes:
            return np.zeros(1)
        return self._ds["lon"].values

    @property
    def lat(self):
        if "Y" not in self.axes:
            return np.zeros(1)
        return self._ds["lat"].values

    @property
    def depth(self):
        if "Z" not in self.axes:
            return np.zeros(1)
        return self._ds["depth"].values

    @property
    def deg2m(self) -> float:
        return self._mesh.deg2m if self._mesh.is_spherical() else 1.0

    @property
    def is_curvilinear(self) -> bool:
        return ("X" in self.axes) and self.lon.ndim == 2

    def get_axis_dim(self, axis: str) -> int:
        """Cells-1 along an axis (xgrid.py:21-24, 220-231): the ravel dims of ``ei``."""
        if axis not in self.axes:
            raise ValueError(f"Axis {axis!r} is not part of this grid. Available axes: {self.axes}")
        fx, fy = self.sgrid_metadata.face_dimensions
        fnp = {"X": fx, "Y": fy}.get(axis)
        if fnp is None:
            fnp = self.sgrid_metadata.vertical_dimensions[0]
        sizes = self._ds.sizes
        if fnp.face in sizes:
            return sizes[fnp.face] - 1
        return _sgrid.get_n_faces(sizes[fnp.node], fnp.padding) - 1

    @property
    def xdim(self):
        return self.get_axis_dim("X")

    @property
    def ydim(self):
        return self.get_axis_dim("Y")

    @property
    def zdim(self):
        return self.get_axis_dim("Z")

    def offsets(self) -> dict:
        """C-grid index offsets from the SGRID padding (_xinterpolators.py:99-109)."""
        md = self.sgrid_metadata
        out = {}
        for fnp, axis in zip(md.face_dimensions, ["X", "Y"]):
            out[axis] = 1 if fnp.padding == _sgrid.Padding.LOW else 0
        if md.vertical_dimensions is not None:
            out["Z"] = 1 if md.vertical_dimensions[0].padding == _sgrid.Padding.LOW else 0
        else:
            out["Z"] = 0
        return out

    def get_axis_dim_mapping(self, dims) -> dict:
        d2a = self.sgrid_metadata.dim_to_axis()
        out = {}
        for dim in dims:
            ax = d2a.get(str(dim))
            if ax in self.axes:
                out[ax] = str(dim)
        return out

    def ravel_index(self, axis_indices: dict) -> np.ndarray:  # basegrid.py:83-118, 254-278
        dims = [self.get_axis_dim(a) for a in self.axes]
        idx = [np.asarray(axis_indices[a], dtype=np.int64) for a in self.axes]
        ei = idx[-1].copy()
        stride = 1
        for i in range(len(dims) - 2, -1, -1):
            stride *= dims[i + 1]
            ei = ei + idx[i] * stride
        return ei

    def unravel_index(self, ei) -> dict:  # basegrid.py:120-152, 219-252
        """The `ei` of a particle column back into one index per axis of this grid (NumPy floor division: negative codes unravel the way
        the reference's do)."""
        dims = np.array([self.get_axis_dim(a) for a in self.axes], dtype=int)
        strides = np.cumprod(dims[::-1])[::-1]
        rest = np.asarray(ei)
        out = np.empty((len(dims), len(rest)), dtype=int)
        for i in range(len(dims) - 1):
            out[i, :] = rest // strides[i + 1]
            rest = rest % strides[i + 1]
        out[-1, :] = rest
        return dict(zip(self.axes, out, strict=True))

    def zonal_periodic(self):  # xgrid.py:294 (a v3 left-over without a body there too)
        return None

    def get_spatial_hash(self) -> SpatialHash:
        if self._spatialhash is None:
            self._spatialhash = SpatialHash(self.lon, self.lat, self._mesh.is_spherical())
        return self._spatialhash
