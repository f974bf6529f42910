"""Trajectory output to Parquet (mirrors src/parcels/_core/particlefile.py).

``ParticleSet.execute(..., output_file=ParticleFile(path, outputdt))`` appends one table per output time holding the
particles with ``|t_p - t| <= |dt|/2`` (particlefile.py:198-221).  The particle columns are the host NumPy SoA dict,
which ``Kernel.execute`` refreshes from the device at every output interval.
"""

from __future__ import annotations

import os
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


def _take_rows(data, names, idx):
    """The to-write columns of the selected rows; when EVERY row is selected (the usual output step of a run whose particles move in
    lock-step) the columns themselves, not 1e7-row copies of them."""
    n = len(data[names[0]]) if names else 0
    if isinstance(idx, np.ndarray) and idx.dtype != bool and len(idx) == n and (n == 0 or (idx[0] == 0 and idx[-1] == n - 1 and np.all(idx[1:] > idx[:-1]))):
        return {k: data[k] for k in names}  # (strictly increasing from 0 to n-1: the identity, not a permutation of it)
    return {k: data[k][idx] for k in names}


def get_schema(pclass, file_metadata, fset_time_interval):
    import pyarrow as pa

    fields = []
    for v in _get_vars_to_write(pclass):
        attrs = {str(k): str(val) for k, val in v.attrs.items()}
        if v.name == "t" and fset_time_interval is not None:
            attrs.update(fset_time_interval.get_cf_attrs())  # particlefile.py:38-41
        fields.append(pa.field(v.name, pa.from_numpy_dtype(np.dtype(v.dtype)), metadata=attrs))
    return pa.schema(fields, metadata={str(k): str(v) for k, v in file_metadata.items()})


class ParticleFile:
    """One Parquet file of trajectories.  With an initialised torch.distributed group of more than one rank (one process per GPU,
    ParticleSet(shard=...)) every rank calls ``write`` collectively: the rows passing the write filter are gathered (RCCL
    all-gather over xGMI under backend "nccl", gloo in the CPU tests -- parcels_amd.distributed.gather_write_columns) and rank 0
    appends the one table; the file is byte-identical to the one a single process writes for the whole id space.
    ``distributed=False`` keeps a ParticleFile rank-local; ``distributed="always"`` takes the collective path even in a group of ONE rank
    (device write filter -> gather to rank 0 -> one table; the preflight of the multi-GPU write-out on a single GPU)."""

    def __init__(self, path, outputdt, compression="zstd", mode=None, distributed=None, group=None, use_dictionary=False, writer="auto",
                 encode_threads=None):
        if not isinstance(outputdt, (np.timedelta64, timedelta, float)):
            raise ValueError(f"Expected outputdt to be a np.timedelta64, datetime.timedelta or float (in seconds), got {type(outputdt)}")
        self._compression = compression
        # dictionary pages are useless for coordinates, times and unique ids and double the encode time (measured: 1.5 s vs 0.7 s
        # per table of 1e7 rows with zstd); the reference leaves pyarrow's default (on) -- same values, same schema either way
        self._use_dictionary = use_dictionary
        # "auto": the multi-threaded writer of parcels_amd/parquet_writer.py for the flat numeric tables a ParticleSet produces (pyarrow
        # encodes a table on one thread: 0.5 s per 1e7 particles), pyarrow's own writer for anything else; "pyarrow" / "fast" force one
        if writer not in ("auto", "pyarrow", "fast"):
            raise ValueError(f"writer must be 'auto', 'pyarrow' or 'fast'. Got {writer!r}")
        self._writer_kind = writer
        self._encode_threads = encode_threads
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
        from .distributed import dist_rank_world

        self._group = group
        self._rank, self._world = dist_rank_world(group) if distributed is not False else (0, 1)
        if distributed not in (None, True, False, "always"):
            raise ValueError(f"distributed must be None, True, False or 'always'. Got {distributed!r}")
        if distributed is True and self._world == 1:
            raise ValueError("ParticleFile(distributed=True) needs an initialised torch.distributed group")
        self._collective = self._world > 1
        if distributed == "always":
            import torch.distributed as dist

            if not (dist.is_available() and dist.is_initialized()):
                raise ValueError("ParticleFile(distributed='always') needs an initialised torch.distributed group (one rank is enough)")
            self._collective = True
        if self._rank == 0:  # the file belongs to rank 0; the other ranks only contribute rows
            if path.exists():
                if mode is None:
                    raise ValueError(f"Path '{path}' already exists. Use mode='w' or use a new path.")
                path.unlink()
            if not path.parent.exists():
                raise ValueError(f"Folder location for '{path} does not exist. Create the folder location first.")
        self.metadata = {}
        self.gather_seconds = 0.0  # time spent in the write-out exchange (multi-rank only)

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
        self.commit_columns(self.prepare_columns(pclass, columns, time_interval))

    def prepare_columns(self, pclass, columns: dict, time_interval=None):
        """First half of write_columns (FastParquetWriter.prepare: the pages start compressing; the file is not touched) -> ticket for
        commit_columns, which must be called in table order.  With pyarrow's writer the ticket just carries the columns."""
        self._ensure_writer(pclass, time_interval)
        names = [v.name for v in _get_vars_to_write(pclass)]
        if hasattr(self._writer, "prepare"):
            return ("fast", self._writer.prepare({n: np.asarray(columns[n]) for n in names}))
        return ("pyarrow", {n: np.asarray(columns[n]) for n in names})

    def commit_columns(self, ticket):
        kind, payload = ticket
        if kind == "fast":
            self._writer.commit(payload)
        else:
            import pyarrow as pa

            self._writer.write_table(pa.table({n: pa.array(a) for n, a in payload.items()}, schema=self._writer.schema))

    def _ensure_writer(self, pclass, time_interval=None):
        import pyarrow.parquet as pq

        if self._writer is None:
            schema = get_schema(pclass, self.metadata, time_interval)
            from .parquet_writer import _CODEC, FastParquetWriter, supports_schema

            fast_ok = supports_schema(schema) and self._compression in _CODEC and not self._use_dictionary
            if self._writer_kind == "fast" and not fast_ok:
                raise ValueError("writer='fast' needs flat bool / int / float Variables, no dictionary pages and compression zstd / lz4 / snappy / gzip / None")
            if fast_ok and self._writer_kind in ("auto", "fast"):
                self._writer = FastParquetWriter(self.path, schema, compression=self._compression, threads=self._encode_threads)
            else:
                self._writer = pq.ParquetWriter(self.path, schema, compression=self._compression, use_dictionary=self._use_dictionary)

    def write(self, pset, t, fieldset=None, indices=None):
        from .columns import readonly

        fieldset = fieldset or pset.fieldset
        if indices is None and getattr(pset, "_prefiltered", False) and not self._collective:
            # a snapshot whose rows already passed the write filter on the device (_AsyncWriter / pk_particles_snapshot_filtered)
            names = [v.name for v in _get_vars_to_write(pset._pclass)]
            self.write_columns(pset._pclass, {k: pset._data[k] for k in names}, fieldset.time_interval)
            return
        data = readonly(pset._data)  # (reads only: a device-resident set downloads the columns touched here and stays clean)
        if isinstance(t, (np.timedelta64, np.datetime64)):
            t = to_seconds(t - fieldset.time_interval.left)
        names = [v.name for v in _get_vars_to_write(pset._pclass)]
        if self._collective:
            import time as _time

            from .distributed import device_write_rows, gather_write_columns

            t0 = _time.perf_counter()
            eng = getattr(fieldset, "_engine", None)
            cols = None
            if indices is None and getattr(pset, "_device_rows_current", False) and eng is not None and self._device_gather_ok(eng, names):
                # the columns are device-resident (ParticleSet.execute): filter there, send the device rows -- no D2H / H2D round trip.
                # Through the C ABI when the library's own communicator is up (pk_gather_rows_to_root: filter, pack, count exchange and
                # row exchange inside libparcels_hip.so, RCCL over xGMI), else the same steps over torch.distributed
                from .distributed import check_abort, ensure_comm

                if ensure_comm(eng, self._group):
                    check_abort()
                    cols = eng.gather_rows(names, float(t))
                    self.gather_seconds += _time.perf_counter() - t0
                    if cols is None:
                        return
                    self.write_columns(pset._pclass, cols, fieldset.time_interval)
                    return
                cols = device_write_rows(eng, names, float(t))
            if cols is None:
                idx = _to_write_particles(data, t) if indices is None else indices  # the reference's filter, applied BEFORE the exchange
                cols = _take_rows(data, names, idx)
            cols = gather_write_columns(cols, self._group, device=getattr(eng, "device", None))
            self.gather_seconds += _time.perf_counter() - t0
            if cols is None:
                return
        else:
            idx = _to_write_particles(data, t) if indices is None else indices
            cols = _take_rows(data, names, idx)
        self.write_columns(pset._pclass, cols, fieldset.time_interval)

    def _device_gather_ok(self, eng, names) -> bool:
        """Every to-write Variable lives in a device column and the exchange runs over RCCL (device tensors)."""
        try:
            import torch.distributed as dist

            from .distributed import _DEVICE_COLUMN_TYPES
        except Exception:
            return False
        spatial = ("z", "y", "x", "dz", "dy", "dx")
        return dist.get_backend(self._group) == "nccl" and all(n in _DEVICE_COLUMN_TYPES or n in spatial for n in names)

    def async_writer(self, pset, engine, out_cols):
        """Writer that takes the output step off the critical path of ParticleSet.execute (None for a collective multi-rank file:
        its all-gather must stay on the thread that drives the launches)."""
        if self._collective:
            return None
        return _AsyncWriter(self, pset, engine, out_cols)

    def close(self):
        if self._writer is not None:
            self._writer.close()
            self.writer_seconds = dict(getattr(self._writer, "seconds", {}) or {})  # FastParquetWriter: waiting for page compression / for the file writes
            self._writer = None

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()


class _SnapshotView:
    """What ParticleFile.write needs of a ParticleSet, over one snapshot of the columns."""

    def __init__(self, data, pclass, fieldset, prefiltered=False):
        self._data, self._pclass, self.fieldset, self._prefiltered = data, pclass, fieldset, prefiltered


class _AsyncWriter:
    """Double-buffered write-out: submit() snapshots the to-write device columns (pk_particles_snapshot_filtered: the write filter and the
    packing in host row order on the device, D2H on the copy stream into one of two pinned column sets) and returns; TWO writer threads
    take the tables: each waits for its copy and starts compressing its pages (FastParquetWriter.prepare), then commits them to the file when
    it is its table's turn -- the compression of table k+1 overlaps the file writes of table k, the tables land in submission order.  At
    most two snapshots are in flight (the third submit waits for the first table)."""

    def __init__(self, pfile, pset, engine, out_cols):
        import threading
        from concurrent.futures import ThreadPoolExecutor

        self.pfile, self.pset, self.engine = pfile, pset, engine
        self.cols = [c for c in out_cols if c in engine._SNAP_COLS or c in engine.device_variables]
        self.pool = ThreadPoolExecutor(max_workers=2, thread_name_prefix="parcels-writeout")
        self.pending = [None, None]
        self.slot = 0
        self.seq = 0                        # tables submitted
        self.turn = 0                       # table whose commit is due
        self.order = threading.Condition()
        self.encode_seconds = 0.0
        self.block_seconds = 0.0
        self.wait_seconds = 0.0  # writer threads: waiting for a snapshot's D2H (the rest of their time is filter + encode + file)

    def submit(self, data, t):
        slot = self.slot
        self.slot ^= 1
        if self.pending[slot] is not None:  # its pinned columns are about to be reused
            import time as _time

            t0 = _time.perf_counter()
            self.pending[slot].result()
            self.block_seconds += _time.perf_counter() - t0  # the launching thread waited for the writer: write-out NOT hidden
        # the write filter runs on the device when every to-write Variable is a device column (the default Particle classes): only the rows
        # of the table cross PCIe and the writer thread encodes them as they are.  With host-only Variables to write, the rows must be chosen
        # on the host, where those live: the full snapshot and NumPy's filter
        pclass = getattr(self.pset, "_pclass", None)
        names = [v.name for v in _get_vars_to_write(pclass)] if pclass is not None else None
        on_device = (names is not None and all(c in self.cols for c in names) and {"t", "dt"} <= set(self.cols)
                     and os.environ.get("PARCELS_AMD_HOST_WRITE_FILTER") != "1")
        if on_device:
            self.engine.snapshot_begin(self.cols, slot, filter_t=float(t))
        else:
            self.engine.snapshot_begin(self.cols, slot)
        # host-only Variables: a device kernel list replaces them (compaction), never mutates them -- the writer thread may read the arrays
        # themselves.  With Python kernels on the host path (hostkernels.execute_hosted: `particles.age += particles.dt` works IN PLACE on
        # these very arrays during the next interval) the table needs its own copy, or it would pair this output time's t / x / y with
        # later values of the Variable.
        mutable = bool(getattr(getattr(self.pset, "_kernel", None), "host_functions", None))
        from .columns import raw_items

        host_only = {k: (np.array(v, copy=True) if mutable else v) for k, v in raw_items(data)
                     if k not in self.engine._SNAP_COLS and k not in self.engine.device_variables}
        seq = self.seq
        self.seq += 1
        self.pending[slot] = self.pool.submit(self._task, slot, host_only, float(t), on_device, seq)

    def _task(self, slot, host_only, t, prefiltered=False, seq=0):
        import time as _time

        ticket = None
        try:
            t0 = _time.perf_counter()
            cols = self.engine.snapshot_wait(slot)
            t1 = _time.perf_counter()
            self.wait_seconds += t1 - t0
            two_phase = prefiltered and hasattr(self.pfile, "prepare_columns") and not getattr(self.pfile, "_collective", False)
            if two_phase:  # (the page compression starts now, whatever the tables in front of this one are doing)
                names = [v.name for v in _get_vars_to_write(self.pset._pclass)]
                ticket = self.pfile.prepare_columns(self.pset._pclass, {k: cols[k] for k in names}, self.pset.fieldset.time_interval)
            with self.order:
                while self.turn != seq:
                    self.order.wait()
            if two_phase:
                self.pfile.commit_columns(ticket)
            else:
                if not prefiltered:
                    cols.update(host_only)
                self.pfile.write(_SnapshotView(cols, self.pset._pclass, self.pset.fieldset, prefiltered=prefiltered), t)
            self.encode_seconds += _time.perf_counter() - t1
        finally:
            with self.order:  # (also when this table failed: the ones behind it must not wait for ever)
                if self.turn == seq:
                    self.turn = seq + 1
                else:  # failed before its turn: wait for it, then pass it on
                    while self.turn != seq:
                        self.order.wait()
                    self.turn = seq + 1
                self.order.notify_all()

    def drain(self):
        for k in (self.slot, self.slot ^ 1):  # oldest first
            if self.pending[k] is not None:
                self.pending[k].result()
                self.pending[k] = None

    def close(self):
        try:
            self.drain()
        finally:
            self.pool.shutdown(wait=True)


def read_particlefile(path):
    """Read a particle file into a pandas DataFrame (particlefile.py:224-286, without time decoding)."""
    import pyarrow.parquet as pq

    return pq.read_table(path).to_pandas()
