"""FieldSet (mirrors src/parcels/_core/fieldset.py): fields, context constants, constant fields."""

from __future__ import annotations

import numpy as np

from .dataset import Dataset
from .field import Field, StructuredModelData, VectorField
from .interpolators import XConstantField, XLinear
from .sgrid import FaceNodePadding, Padding, SGrid2DMetadata

__all__ = ["FieldSet"]


def _constant_field_model(mesh):
    """The 1x1 grid that carries constant fields (model.py:292-317)."""
    ds = Dataset(
        {},
        coords={"lat": (["lat"], np.array([0])), "lon": (["lon"], np.array([0])), "depth": (["depth"], np.array([0])),
                "time": (["time"], np.array([0]))},
        sgrid=SGrid2DMetadata(
            node_dimensions=("lon", "lat"),
            face_dimensions=(FaceNodePadding("XC", "lon", Padding.LOW), FaceNodePadding("YC", "lat", Padding.LOW)),
        ),
    )
    return StructuredModelData(ds, mesh, {})


class FieldSet:
    def __init__(self, models):
        self.__dict__["context"] = None
        self.models = list(models)
        self._fields = None
        self._engine = None
        self._constant_models = {}
        self.reconstruct_fields()
        self.__dict__["context"] = {}

    def __setattr__(self, name, value):
        context = self.__dict__.get("context")
        if context is not None and name in context:
            raise AttributeError(f"Cannot assign '{name}' directly. Use fieldset.context['{name}'] instead.")
        super().__setattr__(name, value)

    @property
    def fields(self):
        if self._fields is None:
            self.reconstruct_fields()
        return self._fields

    def reconstruct_fields(self):
        fields = []
        for model in self.models:
            if model._fields is None:
                model._fields = model.construct_fields()
            else:  # keep interpolator assignments, add fields for new data variables
                have = {f.name for f in model._fields}
                for f in model.construct_fields():
                    if f.name not in have:
                        model._fields.append(f)
            fields += model._fields
        self._fields = {f.name: f for f in fields}
        grids = self.gridset
        for f in self._fields.values():
            f._fieldset = self
            if isinstance(f, Field):
                f.igrid = grids.index(f.grid)
        for f in self._fields.values():
            if isinstance(f, VectorField):
                f.igrid = f.U.igrid
        self._engine = None  # device copies are rebuilt lazily

    def __getattr__(self, name):
        d = self.__dict__
        fields = d.get("_fields") or {}
        if name in fields:
            return fields[name]
        ctx = d.get("context") or {}
        if name in ctx:
            return ctx[name]
        raise AttributeError(f"FieldSet has no attribute '{name}'")

    @property
    def time_interval(self):
        tis = [m.time_interval for m in self.models if m.time_interval is not None]
        if not tis:
            return None
        overlap = tis[0]
        for ti in tis[1:]:
            if overlap is None:
                return None
            overlap = overlap.intersection(ti)
        return overlap

    @property
    def gridset(self):
        grids = []
        for f in self._fields.values():
            if f.grid not in grids:
                grids.append(f.grid)
        return grids

    def add_constant_field(self, name: str, value, mesh="spherical"):
        """fieldset.py:175-205"""
        if mesh not in ("flat", "spherical"):
            raise ValueError(f"mesh must be one of ['flat', 'spherical']. Got {mesh!r}.")
        model = self._constant_models.get(mesh)
        if model is None:
            model = self._constant_models[mesh] = _constant_field_model(mesh)
        from .dataset import DataArray

        model.data.data_vars[name] = DataArray(("mockT", "mockZ", "mockY", "mockX"), np.full((1, 1, 1, 1), float(value)))
        if model not in self.models:
            self.models.append(model)
        self.reconstruct_fields()
        getattr(self, name).interp_method = XConstantField()

    def add_context(self, name, value):
        """fieldset.py:207-222"""
        if not isinstance(name, str) or not name.isidentifier():
            raise ValueError(f"context name has to be a valid Python variable name. Got {name!r}")
        if name in self.context:
            raise ValueError(f"FieldSet already has a context with name '{name}'")
        self.context[name] = value

    @classmethod
    def from_sgrid_conventions(cls, ds, mesh=None, vector_fields=None, skip_field_data_validation=False):
        """fieldset.py:277-313 / model.py:202-250. ``ds``: parcels_amd.Dataset (or an SGRID xarray.Dataset)."""
        if not isinstance(ds, Dataset):
            ds = Dataset.from_xarray(ds)
        if mesh is None:  # model.py:387-401
            md = ds.sgrid
            names = md.node_coordinates or ("lon", "lat")
            units = [ds[n].attrs.get("units") for n in names]
            if any(u is None for u in units):
                raise ValueError("mesh not given and the node coordinates carry no 'units' attribute")
            mesh = "spherical" if isinstance(units[0], str) and "degree" in units[0].lower() else "flat"
        if vector_fields is None:  # model.py:404-412
            vector_fields = {}
            names = set(ds.data_vars)
            if {"U", "V"} <= names:
                vector_fields["UV"] = ("U", "V")
            if {"U", "V", "W"} <= names:
                vector_fields["UVW"] = ("U", "V", "W")
        if not isinstance(vector_fields, dict):
            raise ValueError(f"vector_fields must be a dictionary. Got {type(vector_fields)=!r}.")
        for vname, comps in vector_fields.items():
            if not (2 <= len(comps) <= 3):
                raise ValueError(f"Vector field {vname} must have 2 or 3 components")
            for c in comps:
                if c not in ds.data_vars:
                    raise ValueError(f"Field component '{c}' not present in the source dataset")
        model = StructuredModelData(ds, mesh, vector_fields, skip_field_data_validation)
        model._fields = model.construct_fields()
        for f in model._fields:
            if isinstance(f, Field):
                f.interp_method = XLinear()
        return cls([model])

    # -- device side -------------------------------------------------------------------------------------------
    def _engine_or_create(self, device: int | None = None):
        from .engine import DeviceEngine

        if self._engine is None or (device is not None and self._engine.device != device):
            self.__dict__["_engine"] = DeviceEngine(self, device=0 if device is None else device)
        return self._engine

    def to_device(self, device: int = 0, nslots: int | None = None):
        """Create the device copy now (grids, hash tables, field-level rings).  ``nslots`` bounds the number of
        device-resident time levels per field (the analogue of FieldSet.to_windowed_arrays, fieldset.py:142-173)."""
        from .engine import DeviceEngine

        self.__dict__["_engine"] = DeviceEngine(self, device=device, nslots=nslots)
        return self
