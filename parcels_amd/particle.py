"""Particle schema and SoA storage (mirrors src/parcels/_core/particle.py)."""

from __future__ import annotations

import numpy as np

from .statuscodes import StatusCode

__all__ = ["Particle", "ParticleClass", "Variable", "get_default_particle"]


class Variable:  # particle.py:20-76
    def __init__(self, name, dtype=np.float32, initial=0, to_write=True, attrs=None):
        # the reference's checks, in its order (particle.py:36-60, utils/string.py:4-16)
        if not isinstance(name, str):
            raise TypeError(f"Expected a string for variable name, got {type(name).__name__} instead.")
        if not name.isidentifier():
            raise ValueError(f"Received invalid Python variable name {name!r}: not a valid identifier. HINT: avoid using spaces, special characters, "
                             "and starting with a number.")
        import keyword

        if keyword.iskeyword(name):
            raise ValueError(f"Received invalid Python variable name {name!r}: it is a reserved keyword. HINT: avoid using the following names: "
                             f"{', '.join(keyword.kwlist)}")
        try:
            dt = np.dtype(dtype)
        except (TypeError, ValueError) as e:
            raise TypeError(f"Variable dtype must be a valid numpy dtype. Got {dtype=!r}") from e
        if to_write not in (True, False):
            raise ValueError(f"to_write must be one of {[True, False]!r}. Got {to_write=!r}")
        if not to_write and attrs:
            raise ValueError(f"Attributes cannot be set if {to_write=!r}.")
        self.name = name
        self.dtype = dt.type
        self.initial = initial
        self.to_write = to_write
        self.attrs = dict(attrs or {})

    def __repr__(self):
        return f"Variable(name={self.name!r}, dtype={self.dtype.__name__}, initial={self.initial!r}, to_write={self.to_write!r})"


class ParticleClass:  # particle.py:79-113
    def __init__(self, variables):
        if not all(isinstance(v, Variable) for v in variables):
            raise ValueError("All items in variables must be instances of Variable")
        names = [v.name for v in variables]
        if len(set(names)) != len(names):
            raise ValueError("Variable name already exists")
        self.variables = list(variables)

    def add_variable(self, variable):
        if isinstance(variable, Variable):
            variable = [variable]
        for v in variable:
            if not isinstance(v, Variable):
                raise TypeError(f"Expected Variable, got {type(v)}")
        existing = {v.name for v in self.variables}
        for v in variable:
            if v.name in existing:
                raise ValueError(f"Variable name already exists: {v.name}")
        return ParticleClass(self.variables + list(variable))


def get_default_particle(spatial_dtype) -> ParticleClass:  # particle.py:123-175
    if spatial_dtype not in (np.float32, np.float64):
        raise ValueError(f"spatial_dtype must be np.float32 or np.float64. Got {spatial_dtype=!r}")
    return ParticleClass(
        [
            Variable("t", dtype=np.float64, attrs={"standard_name": "time", "units": "seconds", "axis": "T"}),
            Variable("z", dtype=spatial_dtype, attrs={"standard_name": "vertical coordinate", "units": "m", "positive": "down"}),
            Variable("y", dtype=spatial_dtype, attrs={"standard_name": "latitude", "units": "degrees_north", "axis": "Y"}),
            Variable("x", dtype=spatial_dtype, attrs={"standard_name": "longitude", "units": "degrees_east", "axis": "X"}),
            Variable("dz", dtype=spatial_dtype, to_write=False),
            Variable("dy", dtype=spatial_dtype, to_write=False),
            Variable("dx", dtype=spatial_dtype, to_write=False),
            Variable("particle_id", dtype=np.int64, attrs={"long_name": "Unique identifier for each particle", "cf_role": "trajectory_id"}),
            Variable("dt", dtype=np.float64, initial=1.0, to_write=False),
            Variable("state", dtype=np.int32, initial=StatusCode.Evaluate, to_write=False),
        ]
    )


Particle = get_default_particle(np.float32)


def create_particle_data(*, pclass: ParticleClass, nparticles: int, ngrids: int, initial=None) -> dict:
    """particle.py:182-222: SoA dict of NumPy columns, ``ei`` is (n, ngrids) int32."""
    initial = dict(initial or {})
    variables = {v.name: v for v in pclass.variables}
    assert "ei" not in initial, "'ei' is for internal use"
    for name, values in initial.items():
        if name not in variables:
            raise ValueError(f"Variable {name} is not defined in the ParticleClass.")
        values = np.asarray(values)
        if values.shape != (nparticles,):
            raise ValueError(f"Initial value for {name} must have shape ({nparticles},). Got {values.shape=}")
        initial[name] = np.ascontiguousarray(values.astype(variables[name].dtype))
    data = {"ei": np.zeros((nparticles, ngrids), dtype=np.int32), **initial}
    import operator

    for v in variables.values():
        if v.name not in data:
            if isinstance(v.initial, operator.attrgetter):  # particle.py:213-216: Variable("x0", initial=attrgetter("x"))
                data[v.name] = data[v.initial(_AttrName())].copy().astype(v.dtype)
            else:
                data[v.name] = np.full((nparticles,), v.initial, dtype=v.dtype)
    return data


class _AttrName:
    """attrgetter("x")(_AttrName()) == "x" (the reference's _compat._attrgetter_helper)."""

    def __getattr__(self, name):
        return name
