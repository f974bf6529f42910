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
            self.__dict__["_engine"] = DeviceEngine(self, device=0 if device is None else device, nslots=self.__dict__.get("_window_slots"))
        return self._engine

    def to_windowed_arrays(self, *, max_levels: int | None = None):
        """fieldset.py:142-173: keep a rolling window of time levels resident instead of the whole series -- here: in HBM, as a ring per field
        that the copy stream refills behind the clock.  The reference's default window is the two levels a step
        brackets; the ring holds those plus the level being prefetched, so ``max_levels`` below 3 gives 3 slots.  Takes effect when the
        device copy is created (the first execute / to_device); idempotent; returns self."""
        if max_levels is not None and (not isinstance(max_levels, (int, np.integer)) or max_levels < 1):
            raise ValueError(f"max_levels must be a positive integer or None. Got {max_levels!r}")
        self.__dict__["_window_slots"] = 3 if max_levels is None else max(3, int(max_levels))
        if self._engine is not None and getattr(self._engine, "nslots_request", None) != self.__dict__["_window_slots"]:
            self.__dict__["_engine"] = None  # rebuilt with the ring on the next use
        return self

    def describe(self, buf=None) -> None:
        """fieldset.py:315-330: fields, their interpolators and where their data lives, context values, mesh and time interval."""
        import sys

        (sys.stdout if buf is None else buf).write(_describe(self))

    def to_device(self, device: int = 0, nslots: int | None = None):
        """Create the device copy now (grids, hash tables, field-level rings).  ``nslots`` bounds the number of
        device-resident time levels per field (the analogue of FieldSet.to_windowed_arrays, fieldset.py:142-173)."""
        from .engine import DeviceEngine

        if nslots is None:
            nslots = self.__dict__.get("_window_slots")
        self.__dict__["_engine"] = DeviceEngine(self, device=device, nslots=nslots)
        return self


def _describe(fieldset) -> str:
    """The table of FieldSet.describe (_repr_utils.py:193-281): one row per field / vector field / context value, sorted by grid number,
    type and name; "Parcels backend" says where the levels come from and, once the device copy exists, how many of them are resident."""
    grids = fieldset.gridset
    eng = fieldset.__dict__.get("_engine")
    rows = []
    for f in fieldset.fields.values():
        vector = isinstance(f, VectorField)
        grid = (f.U if vector else f).grid
        backend = "-"
        if not vector:
            data = getattr(getattr(f, "data", None), "data", None)
            backend = "NumPy" if isinstance(data, np.ndarray) else type(data).__name__
            if eng is not None and f.name in getattr(eng, "field_nslots", {}):
                ns, nt = eng.field_nslots[f.name], eng.field_host[f.name].shape[0]
                backend += f" -> HBM ({'all ' + str(nt) + ' levels resident' if ns >= nt else 'ring of ' + str(ns) + ' of ' + str(nt) + ' levels'})"
        rows.append((f.name, "VectorField" if vector else "Field", str(grids.index(grid)), repr(f.interp_method), backend))
    for k, v in fieldset.context.items():
        rows.append((k, "Context", "-", repr(v), "-"))
    rows.sort(key=lambda r: (r[2], r[1], r[0]))
    head = ("Name", "Type", "Grid number", "Interp method / value", "Parcels backend")
    width = [max(len(head[i]), *(len(r[i]) for r in rows)) if rows else len(head[i]) for i in range(5)]
    line = lambda r: "| " + " | ".join(c.ljust(w) for c, w in zip(r, width)) + " |"  # noqa: E731
    table = "\n".join([line(head), "|" + "|".join(":" + "-" * (w + 1) for w in width) + "|"] + [line(r) for r in rows])
    ti = fieldset.time_interval
    return f"{table}\n\n\nmesh: {fieldset.models[0].grid._mesh}\ntime interval: {None if ti is None else (ti.left, ti.right)!r}\n"
