"""Trajectory output to Parquet (mirrors src/parcels/_core/particlefile.py).

``ParticleSet.execute(..., output_file=ParticleFile(path, outputdt))`` appends one table per output time holding the
particles with ``|t_p - t| <= |dt|/2`` (particlefile.py:198-221).  The particle columns are the host NumPy SoA dict,
which ``Kernel.execute`` refreshes from the device at every output interval.
"""

from __future__ import annotations

from datetime import timedelta
from pathlib import Path

import numpy as np

from .field import to_seconds

__all__ = ["ParticleFile", "read_particlefile"]


def _get_vars_to_write(pclass):
    return [v for v in pclass.variables if v.to_write is not False]


def _to_write_particles(particle_data, t):
    """particlefile.py:198-221: particles whose time lies within dt/2 of the output time."""
    tp = particle_data["t"]
    dt = particle_data["dt"]
    fin = np.isfinite(tp)
    with np.errstate(invalid="ignore"):
        sel = ((t - np.abs(dt / 2) <= tp) & (t + np.abs(dt / 2) >= tp)) | (np.isnan(dt) & (tp == t))
    return np.where(sel & fin & np.isfinite(particle_data["particle_id"]))[0]


def get_schema(pclass, file_metadata, fset_time_interval):
    import pyarrow as pa

    fields = []
    for v in _get_vars_to_write(pclass):
        attrs = {str(k): str(val) for k, val in v.attrs.items()}
        if v.name == "t" and fset_time_interval is not None:
            attrs["units"] = f"seconds since {fset_time_interval.left}"
        fields.append(pa.field(v.name, pa.from_numpy_dtype(np.dtype(v.dtype)), metadata=attrs))
    return pa.schema(fields, metadata={str(k): str(v) for k, v in file_metadata.items()})


class ParticleFile:
    def __init__(self, path, outputdt, compression="zstd", mode=None):
        if not isinstance(outputdt, (np.timedelta64, timedelta, float)):
            raise ValueError(f"Expected outputdt to be a np.timedelta64, datetime.timedelta or float (in seconds), got {type(outputdt)}")
        self._compression = compression
        outputdt = to_seconds(outputdt)
        path = Path(path)
        if path.suffix != ".parquet":
            raise ValueError(f"ParticleFile data is stored in Parquet files - file extension must be '.parquet'. Got {path.suffix=!r}.")
        if outputdt <= 0:
            raise ValueError(f"outputdt must be positive/non-zero. Got {outputdt=!r}")
        self._outputdt = outputdt
        self._path = path
        self._writer = None
        if mode not in {None, "w"}:
            raise ValueError(f"Invalid mode value {mode!r}. Expected one of None or 'w'.")
        if path.exists():
            if mode is None:
                raise ValueError(f"Path '{path}' already exists. Use mode='w' or use a new path.")
            path.unlink()
        if not path.parent.exists():
            raise ValueError(f"Folder location for '{path} does not exist. Create the folder location first.")
        self.metadata = {}

    def set_metadata(self, parcels_grid_mesh):
        from . import __version__

        self.metadata.update({
            "feature_type": "trajectory",
            "Conventions": "CF-1.6/CF-1.7",
            "ncei_template_version": "NCEI_NetCDF_Trajectory_Template_v2.0",
            "parcels_version": __version__,
            "parcels_grid_mesh": repr(parcels_grid_mesh),
        })

    @property
    def outputdt(self):
        return self._outputdt

    @property
    def path(self):
        return self._path

    def write_columns(self, pclass, columns: dict, time_interval=None):
        """Append one table given ready-made columns (used by the multi-GPU write-out on rank 0)."""
        import pyarrow as pa
        import pyarrow.parquet as pq

        if self._writer is None:
            self._writer = pq.ParquetWriter(self.path, get_schema(pclass, self.metadata, time_interval), compression=self._compression)
        self._writer.write_table(pa.table({v.name: pa.array(np.asarray(columns[v.name])) for v in _get_vars_to_write(pclass)},
                                          schema=self._writer.schema))

    def write(self, pset, t, fieldset=None, indices=None):
        fieldset = fieldset or pset.fieldset
        data = pset._data
        if isinstance(t, (np.timedelta64, np.datetime64)):
            t = to_seconds(t - fieldset.time_interval.left)
        idx = _to_write_particles(data, t) if indices is None else indices
        cols = {v.name: data[v.name][idx] for v in _get_vars_to_write(pset._pclass)}
        self.write_columns(pset._pclass, cols, fieldset.time_interval)

    def close(self):
        if self._writer is not None:
            self._writer.close()
            self._writer = None

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()


def read_particlefile(path):
    """Read a particle file into a pandas DataFrame (particlefile.py:224-286, without time decoding)."""
    import pyarrow.parquet as pq

    return pq.read_table(path).to_pandas()
