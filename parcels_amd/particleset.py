"""ParticleSet: SoA particle container and the outer time loop (mirrors src/parcels/_core/particleset.py)."""

from __future__ import annotations

import datetime
import types
import warnings

import numpy as np

from .columns import LazyColumns
from .field import to_seconds
from .kernel import Kernel
from .particle import Particle, create_particle_data

__all__ = ["ParticleSet"]


class ParticleSetWarning(UserWarning):  # _core/warnings.py:14-17
    pass


def _convert_dt_to_float(dt):  # particleset.py:497-506
    try:
        if isinstance(dt, (datetime.timedelta, np.timedelta64)):
            dt = to_seconds(dt)
        dt = float(dt)
        if not np.isfinite(dt) or dt == 0:
            raise ValueError("zero or non-finite")
    except (ValueError, TypeError) as e:
        raise ValueError(f"dt must be a non-zero datetime.timedelta or np.timedelta64 object, got {dt=!r}") from e
    return dt, (1 if dt > 0 else -1)


def _warn_outputdt_release_desync(outputdt, starttime, release_times):  # particleset.py:473-482
    if outputdt and np.isfinite(outputdt):
        t = np.asarray(release_times, dtype=np.float64)
        fin = np.isfinite(t)
        if np.any(np.mod(t[fin] - starttime, outputdt) != 0):
            warnings.warn(
                "Some of the particles have a start time difference that is not a multiple of outputdt. "
                "This could cause the first output of some of the particles that start later "
                "in the simulation to be at a different time than expected.",
                ParticleSetWarning, stacklevel=3)


def _warn_particle_times_outside_fieldset_time_bounds(release_times, time_interval):  # particleset.py:485-494
    t = np.asarray(release_times, dtype=np.float64)
    if t.size == 0 or np.isnan(t).all():
        return
    if np.any(t < 0) or np.any(t > time_interval.time_length_as_flt):
        warnings.warn("Some particles are set to be released outside the FieldSet's executable time domain.", ParticleSetWarning, stacklevel=3)


class ParticleSetView:
    """Attribute access to one row (or an index selection) of the SoA columns."""

    def __init__(self, data, index):
        object.__setattr__(self, "_data", data)
        object.__setattr__(self, "_index", index)

    def __getattr__(self, name):
        try:
            return self._data[name][self._index]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        if name not in self._data:
            raise AttributeError(f"particles have no Variable {name!r}")
        self._data[name][self._index] = value


class ParticleSet:
    """Collection of particles stored as a dict of NumPy columns (particle.py:182-222).

    ``seed`` keys the counter-based RNG of the stochastic kernels; ``sort_by_cell`` lets the engine reorder the
    device copy by grid cell for gather locality (host row order is never affected): "auto" (default) sorts sets of at least
    ``SORT_AUTO_MIN`` particles; ``resort_every`` (seconds of model time, None = ``RESORT_EVERY_DEFAULT``, 0 = never) is the cadence at
    which a long fused launch is cut and re-sorted (measured on BASELINE config 2)."""

    SORT_AUTO_MIN = 100_000
    # The columns stay in HBM between execute() calls and the host arrays are a lazy mirror (parcels_amd/columns.py) at EVERY size: the mirror
    # hands out one persistent ndarray per column and follows references held outside the set (refreshed in place after a launch, uploaded
    # before the next), so a script that keeps `x = pset.x` across calls, or writes through it, sees what it sees with the reference.
    # RESIDENT_MIN is only a default one can raise: sets smaller than it use the eager protocol of rounds 1-4 (every call uploads and
    # downloads everything).  `pset.resident_columns = True / False` forces either.
    RESIDENT_MIN = 0
    RESORT_EVERY_DEFAULT = 30 * 86400.0  # profiles/r03_c2_long_run.json: over 23 days of C2 the locality of ONE sort does not decay (re-sorting only costs)


    def __init__(self, fieldset, pclass=Particle, *, t=None, z=None, y=None, x=None, particle_ids=None, seed=0,
                 sort_by_cell="auto", resort_every=None, shard=None, **kwargs):
        """``shard``: None = all particles live here; ``(rank, world)`` or ``"auto"`` (rank / world size of the initialised
        torch.distributed group) = every process is given the SAME arrays and keeps its contiguous block of the id space
        (parcels_amd.distributed.shard_slice) -- fields are replicated, particles never migrate, and a ParticleFile gathers
        the to-write columns of all ranks at every output time."""
        object.__setattr__(self, "_data", None)
        object.__setattr__(self, "_shard", None)
        self.fieldset = fieldset
        self._kernel = None
        self.seed = int(seed)
        self._sort_by_cell = sort_by_cell if isinstance(sort_by_cell, str) else bool(sort_by_cell)
        if isinstance(sort_by_cell, str) and sort_by_cell != "auto":
            raise ValueError(f"sort_by_cell must be True, False or 'auto'. Got {sort_by_cell!r}")
        self.resort_every = resort_every
        self.resident_columns = "auto"  # see RESIDENT_MIN
        self.device_compaction = True  # deleted particles are removed on the device (False: through NumPy on the host)
        self.async_output = True  # ParticleFile tables are encoded on a writer thread behind the next interval (False: inline)
        self._last_stats = None
        self._t_live = None
        t = np.empty(shape=0) if t is None else np.array(t).flatten()
        y = np.empty(shape=0) if y is None else np.array(y).flatten()
        x = np.empty(shape=0) if x is None else np.array(x).flatten()
        if particle_ids is None:
            particle_ids = np.arange(x.size)
        if z is None:  # particleset.py:82-93
            minz = None
            for field in self.fieldset.fields.values():
                for depth in field.grid.depth:
                    if minz is None or np.abs(depth) < np.abs(minz):
                        minz = depth
            z = np.ones(x.size) * minz if minz is not None else np.zeros(x.size)
        else:
            z = np.array(z).flatten()
            if z.size == 1 and x.size > 1:
                z = np.repeat(z, x.size)
        assert x.size == y.size and x.size == z.size, "x, y, z don't all have the same lengths"
        if t is None or len(t) == 0:
            t = np.array(np.nan)
        elif isinstance(t[0], np.datetime64) and self.fieldset.time_interval:
            t = to_seconds(t - self.fieldset.time_interval.left)
        elif isinstance(t[0], np.timedelta64):
            t = to_seconds(t)
        elif np.issubdtype(np.asarray(t).dtype, np.floating) or np.issubdtype(np.asarray(t).dtype, np.integer):
            t = np.asarray(t, dtype=np.float64)  # seconds since fieldset.time_interval.left
        else:
            raise TypeError("particle t must be a datetime, timedelta, or float seconds")
        t = np.repeat(t, x.size) if np.size(t) == 1 else np.asarray(t)
        assert x.size == t.size, "t and positions (x, y, z) do not have the same lengths."
        if self.fieldset.time_interval:  # particleset.py:108-110
            _warn_particle_times_outside_fieldset_time_bounds(t, self.fieldset.time_interval)
        for kwvar in kwargs:
            kwargs[kwvar] = np.array(kwargs[kwvar]).flatten()
            assert x.size == kwargs[kwvar].size, f"{kwvar} and positions (x, y, z) don't have the same lengths."
        if shard is not None:
            from .distributed import resolve_shard, shard_slice

            rank, world = resolve_shard(shard)
            sl = shard_slice(x.size, rank, world)
            particle_ids = np.asarray(particle_ids)[sl]
            t, z, y, x = t[sl], z[sl], y[sl], x[sl]
            kwargs = {k: v[sl] for k, v in kwargs.items()}
            self._shard = (rank, world)
        self._data = create_particle_data(
            pclass=pclass, nparticles=x.size, ngrids=len(fieldset.gridset),
            initial=dict(t=t, z=z, y=y, x=x, particle_id=np.asarray(particle_ids)),
        )
        self._pclass = pclass
        names = [v.name for v in pclass.variables]
        for kwvar, kwval in kwargs.items():
            if kwvar not in names:
                raise RuntimeError(f"Particle class does not have Variable {kwvar}")
            self._data[kwvar][:] = kwval

    @property
    def sort_by_cell(self) -> bool:
        """Whether launches cell-sort the device copy ("auto": from SORT_AUTO_MIN particles on)."""
        if isinstance(self._sort_by_cell, str):
            return len(self) >= self.SORT_AUTO_MIN
        return self._sort_by_cell

    @sort_by_cell.setter
    def sort_by_cell(self, value):
        self._sort_by_cell = value if isinstance(value, str) else bool(value)

    # -- container protocol (particleset.py:140-190) ---------------------------------------------------------------
    def __getattr__(self, name):
        data = self.__dict__.get("_data")
        if data is not None and name in data:
            return data[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        data = self.__dict__.get("_data")
        if name != "_data" and isinstance(data, (dict, LazyColumns)) and name in data:
            data[name][:] = value
        else:
            if name == "_data" and isinstance(value, dict) and not isinstance(value, LazyColumns):
                value = LazyColumns(value)  # the host mirror of columns that may live on the device (parcels_amd/columns.py)
            object.__setattr__(self, name, value)

    def __len__(self):
        d = self._data
        return len(d.raw("particle_id") if isinstance(d, LazyColumns) else d["particle_id"])  # (the length is never stale)

    def __iter__(self):  # particleset.py:143-153
        self._index = 0
        return self

    def __next__(self):
        if self._index < len(self):
            p = self[self._index]
            self._index += 1
            return p
        raise StopIteration

    def __getitem__(self, index):
        """One particle (or a sub-selection) by index: attribute reads and writes go to the set's columns (the part of the
        reference's ParticleSetView, particlesetview.py:17-96, that user code outside kernels relies on)."""
        return ParticleSetView(self._data, index)

    def add(self, particles):  # particleset.py:186-225
        """Append the particles of another ParticleSet; their ids continue after the largest id of this set."""
        assert particles is not None, f"Trying to add another {type(self)} to this one, but the other one is None - invalid operation."
        assert type(particles) is type(self)
        if len(particles) == 0:
            return
        if len(self) == 0:
            self._data = particles._data
            return
        offset = self._data["particle_id"].max() + 1
        particles._data["particle_id"] = particles._data["particle_id"] + offset
        for d in self._data:
            self._data[d] = np.concatenate((self._data[d], particles._data[d]))
        return self

    def __iadd__(self, particles):
        self.add(particles)
        return self

    @property
    def size(self):
        return len(self)

    @classmethod
    def from_particlefile(cls, fieldset, pclass, filename, restart=True, restarttime=None, **kwargs):  # particleset.py:264-292
        raise NotImplementedError("ParticleSet.from_particlefile is not yet implemented in v4.")

    def data_indices(self, variable_name, compare_values, invert=False):
        """Rows whose `variable_name` equals (one of) `compare_values` -- or does not, with invert (particleset.py:294-319)."""
        compare_values = np.array([compare_values]) if type(compare_values) not in [list, dict, np.ndarray] else compare_values
        return np.where(np.isin(self._data[variable_name], compare_values, invert=invert))[0]

    def set_variable_write_status(self, var, write_status):
        """Whether the Variable `var` is written by a ParticleFile (particleset.py:343-353).  The particle class is shared between
        sets: this set gets its own copy with the changed Variable."""
        from .particle import ParticleClass, Variable

        if not any(v.name == var for v in self._pclass.variables):
            raise KeyError(f"particles have no Variable {var!r}")
        changed = [Variable(v.name, dtype=v.dtype, initial=v.initial, to_write=write_status, attrs=(v.attrs if write_status else None))
                   if v.name == var else v for v in self._pclass.variables]
        self._pclass = ParticleClass(changed)

    def remove_indices(self, indices):  # particleset.py:247-250
        for d in self._data:
            self._data[d] = np.delete(self._data[d], indices, axis=0)

    def _engine(self):
        return self.fieldset._engine_or_create()

    def populate_indices(self):
        """Pre-populate the ``ei`` guesses (particleset.py:252-262) with a device search (pk_search)."""
        eng = self._engine()
        for i in range(len(self.fieldset.gridset)):
            self._data["ei"][:, i] = eng.search(i, self._data["z"], self._data["y"], self._data["x"])

    # -- the outer time loop (particleset.py:355-470) ------------------------------------------------------------
    def execute(self, kernels, dt, endtime=None, runtime=None, output_file=None, verbose_progress=False):
        # A multi-rank ParticleFile makes write() a collective: every rank must then make the SAME sequence of write() calls at the
        # SAME output times, whatever its own shard looks like (empty from the start, emptied by deletions, later releases).
        collective = output_file is not None and bool(getattr(output_file, "_collective", getattr(output_file, "_world", 1) > 1))
        if len(self) == 0 and not collective:
            return
        if isinstance(kernels, types.FunctionType):
            kernels = [kernels]
        self._kernel = Kernel(kernels, self)
        dt, sign_dt = _convert_dt_to_float(dt)
        # Columns that are still device-resident from the previous call stay there (parcels_amd/columns.py): `particles.dt = dt` and the
        # reductions over the release times below run on the device, nothing crosses PCIe unless the host touched the set in between.
        data = self._data
        want_resident = self.resident_columns if isinstance(self.resident_columns, bool) else len(self) >= self.RESIDENT_MIN
        eng0 = data._engine if want_resident and isinstance(data, LazyColumns) and data.resident() and hasattr(data._engine, "fill_column") else None
        if eng0 is not None and eng0 is not getattr(self.fieldset, "_engine", None):
            eng0 = None  # (resident on an engine the FieldSet no longer uses: the host path below downloads what it touches)
        if eng0 is not None:
            eng0.fill_column("dt", dt)
        else:
            self._data["dt"][:] = dt
        t_on_device = eng0 is not None and "t" in data._stale
        t_lo = t_hi = None
        t_nan = 0
        if t_on_device and len(self) > 0:
            t_lo, t_hi, t_nan = eng0.t_stats()
        t_ro = (lambda: data.peek("t")) if isinstance(data, LazyColumns) else (lambda: self._data["t"])  # read-only look at the release times
        if runtime is not None:  # _convert_runtime_to_float (particleset.py:508-520)
            try:
                runtime = to_seconds(runtime) if isinstance(runtime, (datetime.timedelta, np.timedelta64)) else float(runtime)
            except (ValueError, TypeError) as e:
                raise ValueError(f"The runtime must be a datetime.timedelta, np.timedelta64 or float object. Got {type(runtime)}") from e
            if runtime < 0:
                raise ValueError(f"The runtime must be a non-negative timedelta or float. Got {runtime=!r}")
        first = None
        if collective:
            # the first release over ALL shards (NaN-propagating like rel.min(): one unset release time anywhere => fieldset start)
            from .distributed import allreduce_scalars, clear_abort

            clear_abort()  # (a note this rank left in the store when an earlier call failed)
            if t_lo is not None:
                mine = np.nan if t_nan else (t_lo if sign_dt == 1 else t_hi)
            else:
                rel = t_ro()
                mine = (rel.min() if sign_dt == 1 else rel.max()) if len(rel) else np.nan
            unset = bool(len(self)) and bool(np.isnan(mine))
            lo_hi = allreduce_scalars([sign_dt * mine if np.isfinite(mine) else np.inf, -1.0 if unset else -0.0], "min", output_file._group,
                                      device=getattr(getattr(self.fieldset, "_engine", None), "device", None))
            first = np.nan if (lo_hi[1] < 0 or not np.isfinite(lo_hi[0])) else sign_dt * lo_hi[0]
        if first is None and t_lo is not None:
            first = np.nan if t_nan else (t_lo if sign_dt == 1 else t_hi)  # NaN-propagating like rel.min() (one unset release time => fieldset start)
        start_time, end_time = self._start_and_end_times(runtime, endtime, sign_dt, first=first)
        if (t_nan > 0) if t_lo is not None else bool(np.isnan(t_ro()).any()):
            self._data["t"][:] = start_time
        outputdt = output_file.outputdt if output_file else None
        if outputdt and np.isfinite(outputdt):
            _warn_outputdt_release_desync(outputdt, start_time, t_ro())
        next_output = None
        if output_file:
            output_file.set_metadata(self.fieldset.gridset[0]._mesh)
            output_file.metadata["parcels_kernels"] = self._kernel.funcname
            output_file.write(self, start_time)
            next_output = start_time + outputdt * sign_dt
        time = start_time
        from contextlib import nullcontext

        # The particle columns stay DEVICE-RESIDENT for the whole call: one upload here, one download at the end; an
        # output interval copies back only the columns the ParticleFile writes.  (The reference re-reads every column
        # through ParticleSetView on every step, particlesetview.py:97-301.)
        engine = self._engine()
        kern = self._kernel
        # (= the first release: what Kernel.launch would reduce from `t`.  In a collective run `start_time` is the first release over ALL
        # shards; this shard's own may be later -- its launches then start from ITS first release, not from a level it never samples)
        self._t_live = start_time if np.isfinite(start_time) else None
        if collective and self._t_live is not None and len(self) > 0:
            if t_lo is not None:
                own = np.nan if t_nan else (t_lo if sign_dt == 1 else t_hi)
            else:
                rel = t_ro()
                own = rel.min() if sign_dt == 1 else rel.max()
            if np.isfinite(own):
                self._t_live = float(own)
        if kern.host_functions and not kern._jit_tried:
            kern._try_jit(self)  # elementwise Python kernels are compiled into the device program here (parcels_amd/jit.py)
        engine.device_variables = list(kern.device_variables)  # user Variables that device kernels write live on the device
        lazy = want_resident and isinstance(self._data, LazyColumns) and hasattr(engine, "attach")  # (small sets, stand-in engines of the CPU suite: the eager path)
        if len(self) > 0:
            if lazy:
                engine.attach(self._data)  # uploads what the host touched since the last launch (everything the first time)
            else:
                engine.bind_particles(self._data)
                engine.h2d()
        from .columns import readonly

        have_guess0 = kern._have_guess0(readonly(self._data)) if len(self) > 0 else 1
        out_cols = None
        writer = None
        if output_file is not None:
            out_cols = sorted({v.name for v in self._pclass.variables if v.to_write is not False} | {"t", "dt", "state", "particle_id"})
            # the file's own rank-local write path can run behind the next interval (ParticleFile.async_writer); a duck-typed or
            # collective (multi-rank) file is written synchronously
            make = getattr(output_file, "async_writer", None)
            writer = make(self, engine, out_cols) if make is not None and self.async_output else None
        synced = True
        if collective and not kern.host_functions and hasattr(engine, "execute_idle"):
            # One batch over all shards (DeviceEngine.execute): the ranks agree on the first erring iteration and on the first sample outside
            # a field's time interval of every Kernel.execute, and on the error codes it ends with.  (The schedule of a collective run is
            # lock-step -- every rank makes every interval -- which is what lets an agreement sit inside it.)
            from .distributed import batch_agreement

            engine.agree_min, engine.agree_codes = batch_agreement(output_file._group, engine.device, engine=engine)
        try:
            with output_file if output_file is not None else nullcontext():  # the Parquet footer is written on error too
                try:
                    if collective:
                        from .distributed import CollectiveAbort, post_abort
                    while sign_dt * (time - end_time) < 0:
                        if next_output is not None:
                            next_time = (min if sign_dt > 0 else max)(next_output, end_time)
                        else:
                            next_time = end_time
                        stats = None
                        if len(self) == 0 and getattr(engine, "agree_min", None) is not None:
                            stats = engine.execute_idle()  # (an empty shard of a collective run keeps the schedule of agreements and of write())
                        if len(self) > 0:
                            stats = kern.launch(self, next_time, dt, have_guess0=have_guess0)
                            have_guess0 = 1
                            synced = False
                            if lazy and not kern.host_functions:
                                engine.mark_launched(self._data)  # the host arrays are stale now; whoever reads one downloads it
                            self._t_live = next_time if not np.isnan(next_time) else None
                            if kern.only_deletions(stats) and self.device_compaction:
                                # Kernel.remove_deleted on the device: the columns do not leave HBM (pk_particles_compact)
                                self._data = engine.compact_deleted(self._data)
                                synced = len(self) == 0
                            elif kern.needs_host_pass(stats):
                                if lazy:
                                    self._data.sync()
                                else:
                                    engine.d2h()
                                synced = True
                                shard_codes = stats.get("codes_any_shard") if stats is not None else None
                                kern.finish_on_host(self, first_code=shard_codes[0] if shard_codes else None)  # compacts / raises
                                if len(self) > 0:
                                    if lazy:
                                        engine.attach(self._data)
                                    else:
                                        engine.bind_particles(self._data)
                                        engine.h2d()
                        if stats is not None and stats.get("codes_any_shard"):
                            # a particle of ANOTHER shard ended the call in an error state: the reference raises for the batch (kernel.py:
                            # 236-245) -- this rank had nothing to raise above, so it raises the same exception here
                            from .statuscodes import ErrorsToThrow, StatusCode

                            code = int(stats["codes_any_shard"][0])  # (StatusCode is a table of integers, not an enum)
                            empty = np.empty(0)
                            if not synced and len(self) > 0:
                                if lazy:
                                    self._data.sync()
                                else:
                                    engine.d2h()
                                synced = True
                            if code == StatusCode.ErrorOutsideTimeInterval:
                                ErrorsToThrow[code](empty)
                            else:
                                ErrorsToThrow[code](empty, empty, empty)
                        if collective:  # the reference's `if len(pset) == 0: break`, decided over all shards
                            from .distributed import allreduce_scalars

                            # (with an output file the loop goes on over the emptied set like the single-process one below -- every rank
                            # knows `next_output`, so the same row groups / observation times come out on 1 GPU and on N)
                            if next_output is None and allreduce_scalars([float(len(self))], "sum", output_file._group, device=engine.device)[0] == 0:
                                break
                        elif len(self) == 0 and next_output is None:
                            break  # (with an output file the reference's loop goes on to the end time and writes an (empty) table at every
                            #         remaining output time, particleset.py:444-462: same calls here)
                        if next_output is not None and np.abs(next_time - next_output) < 0.001:
                            if writer is not None and not synced:
                                # snapshot on the device, D2H on the copy stream, filter + Parquet encode on the writer thread --
                                # all of it overlaps the next interval's launch (particlefile.py:142-180 off the critical path)
                                writer.submit(self._data, next_output)
                            else:
                                if writer is not None:
                                    writer.drain()  # keep the tables in time order
                                names = [v.name for v in self._pclass.variables if v.to_write is not False]
                                probe = getattr(output_file, "_device_gather_ok", None)
                                on_device = bool(collective and not synced and len(self) > 0 and probe is not None and probe(engine, names))
                                if not synced and not on_device:
                                    if lazy:
                                        self._data.sync(out_cols)
                                    else:
                                        engine.d2h(out_cols)
                                self._device_rows_current = on_device  # a collective file filters and gathers the DEVICE rows
                                try:
                                    output_file.write(self, next_output)
                                finally:
                                    self._device_rows_current = False
                            if np.isfinite(outputdt):
                                next_output += outputdt * sign_dt
                        time = next_time
                except BaseException as e:
                    # a rank-local failure anywhere in an interval of a collective run (restore, key exchange, write, compaction): tell the
                    # others through the store before they wait for this rank in their next collective (distributed.post_abort)
                    if collective and not isinstance(e, CollectiveAbort):
                        post_abort(e)
                    raise
                finally:
                    # pending tables are encoded BEFORE the file is closed (also when a kernel raised: the footer then covers
                    # every table submitted so far)
                    if writer is not None:
                        try:
                            writer.close()
                        finally:  # (where the write-out went: bench.py / tools/bench_writeout.py print it)
                            self._last_writer_stats = {"snapshot_wait_s": getattr(writer, "wait_seconds", None), "filter_encode_file_s": getattr(writer, "encode_seconds", None),
                                                       "submit_block_s": getattr(writer, "block_seconds", None)}
        finally:
            if getattr(engine, "agree_min", None) is not None:
                self._agreement_stats = dict(getattr(engine.agree_min, "stats", {}) or {})  # calls / seconds inside the batch agreements of this execute
            engine.agree_min = engine.agree_codes = None
            if not synced and len(self) > 0 and not lazy:
                engine.d2h()
            # (lazy: the columns stay on the device; the host arrays are refreshed column by column when somebody reads them, and the
            # next execute() uploads only what the host touched -- parcels_amd/columns.py)
            self._t_live = None

    def _start_and_end_times(self, runtime, endtime, sign_dt, first=None):  # particleset.py:523-585
        ti = self.fieldset.time_interval
        if runtime is not None and endtime is not None:
            raise ValueError(f"runtime and endtime are mutually exclusive - provide one or the other. Got {runtime=!r}, {endtime=!r}")
        if runtime is None and ti is None:
            raise ValueError("The runtime must be provided when the time_interval is not defined for a fieldset.")
        if runtime is None and endtime is None:
            raise ValueError("Either runtime or endtime must be provided.")
        if first is None:  # (given: the first release over all shards of a collective run, or reduced on the device)
            d = self._data
            rel = d.peek("t") if isinstance(d, LazyColumns) else d["t"]
            first = rel.min() if sign_dt == 1 else rel.max()  # NaN-propagating like the reference: one unset release time => fieldset start
        if ti is not None and endtime is not None:
            if type(endtime) != type(ti.left):  # noqa: E721
                raise ValueError(f"The endtime must be of the same type as the fieldset.time_interval start time. Got {endtime=!r} with time_interval={ti!r}")
            if endtime not in ti:
                raise ValueError(
                    f"Calculated/provided end time of {endtime!r} is not in fieldset time interval {ti!r}. Either reduce your runtime, modify your "
                    "provided endtime, or change your release timing."
                    "Important info:\n"
                    f"    First particle release: {first!r}\n"
                    f"    runtime: {runtime!r}\n"
                    f"    (calculated) endtime: {endtime!r}")
            endtime = to_seconds(endtime - ti.left)
        if sign_dt == 1:
            fieldset_start = 0.0
        else:
            fieldset_start = ti.time_length_as_flt if ti is not None else float(runtime)
        start_time = first if not np.isnan(first) else fieldset_start
        if endtime is None:
            endtime = start_time + sign_dt * float(runtime)
        return float(start_time), float(endtime)
