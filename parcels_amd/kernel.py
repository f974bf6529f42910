"""Kernel: validation of the kernel list and the device replacement of ``Kernel.execute``
(mirrors src/parcels/_core/kernel.py).

The reference runs every kernel function on a NumPy view of the particle set, one dt at a time
(kernel.py:188-245).  Here the kernel list is translated to PK_KERNEL_* ids and ONE ``pk_execute`` call advances all
particles to ``endtime`` on the GPU.  User-written Python kernels join that launch when they are elementwise (parcels_amd/jit.py
translates and compiles them into the kernel-list interpreter); any other Python function makes the loop run on the host columns
(parcels_amd/hostkernels.py) with the built-in kernels' bodies still on the GPU.
"""

from __future__ import annotations

import inspect
import types
import warnings

import numpy as np

from . import kernels as _k
from .engine import raise_particle_errors
from .statuscodes import StatusCode


_RESERVED_COLUMNS = {"t", "z", "y", "x", "dz", "dy", "dx", "dt", "next_dt", "state", "ei", "particle_id"}


class KernelWarning(RuntimeWarning):
    pass


def _same_signature(f, ref):  # _python.py:31-49
    pf, pr = inspect.signature(f).parameters, inspect.signature(ref).parameters
    return [(p.name, p.kind) for p in pf.values()] == [(p.name, p.kind) for p in pr.values()]


class Kernel:
    def __init__(self, kernels, pset):
        if not isinstance(kernels, list):
            raise ValueError(f"kernels must be a list. Got {kernels=!r}")
        for f in kernels:
            if not isinstance(f, types.FunctionType):
                raise TypeError(f"Argument `kernels` should be a function or list of functions. Got {type(f)}")
            if not _same_signature(f, _k.AdvectionRK4):
                raise ValueError(f"Kernel function {f.__name__} must have the signature (particles, fieldset)")
        if len(kernels) == 0:
            raise ValueError("List of `kernels` should have at least one function.")
        # Python functions that are not built-in device kernels: the loop of Kernel.execute then runs on the host columns and only
        # the built-in kernels' bodies on the GPU (parcels_amd/hostkernels.py) -- correct, and slow by construction
        self.host_functions = [f.__name__ for f in kernels if _k.kernel_id(f) is None]
        self.user_program = None   # parcels_amd.jit.UserProgram once the Python functions of the list were compiled for the device
        self.jit_report = None     # why they were not (the host path runs them then)
        self._jit_tried = False
        self._fieldset = pset.fieldset
        self._pclass = pset._pclass
        for f in kernels:
            self.check_fieldsets_in_kernels(f)
        self._kernels = kernels
        self.kernel_ids = [_k.kernel_id(f) for f in kernels]
        # SampleField tokens: the sampled scalar field and the particle Variable that receives the value, per kernel-list slot;
        # those Variables become device columns (pk_particles_desc.extra), every other user Variable stays on the host
        self.samples = {}
        self.device_variables = []
        names = {v.name: v for v in self._pclass.variables}
        for slot, f in enumerate(kernels):
            spec = getattr(f, "_pk_sample", None)
            if spec is None:
                continue
            fname, vname = spec
            fld = self._fieldset.fields.get(fname)
            vector = isinstance(vname, tuple)
            if vector:  # particles.a, particles.b[, particles.c] = fieldset.UV[W][particles]
                if fname not in ("UV", "UVW") or fld is None or not hasattr(fld, "U"):
                    raise ValueError(f"SampleField: a tuple of Variables samples the vector field 'UV' or 'UVW', got '{fname}'")
                if len(vname) != len(fname):
                    raise ValueError(f"SampleField: fieldset.{fname}[particles] returns {len(fname)} components, `into` names {len(vname)}")
            elif fld is None or hasattr(fld, "U"):
                raise ValueError(f"SampleField: '{fname}' is not a scalar field of the fieldset")
            cols = []
            for vn in (vname if vector else (vname,)):
                if vn is None:
                    cols.append(0xFF)  # PK_SAMPLE_DISCARD
                    continue
                if vn not in names or vn in _RESERVED_COLUMNS:
                    raise ValueError(f"SampleField: the ParticleClass has no user Variable '{vn}' (Particle.add_variable)")
                if np.dtype(names[vn].dtype) not in (np.dtype(np.float32), np.dtype(np.float64)):
                    raise TypeError(f"SampleField: Variable '{vn}' must be float32 or float64")
                if vn not in self.device_variables:
                    self.device_variables.append(vn)
                cols.append(self.device_variables.index(vn))
            if fname in ("U", "V", "W"):  # field.py:187-190
                warnings.warn("Sampling of velocities should normally be done using fieldset.UV or fieldset.UVW object; tread carefully",
                              RuntimeWarning, stacklevel=3)
            self.samples[slot] = (fname, sum(c << (8 * j) for j, c in enumerate(cols)) if vector else cols[0])
        from . import _hip as _h

        if len(self.device_variables) > _h.PK_MAX_EXTRA:
            raise ValueError(f"at most {_h.PK_MAX_EXTRA} particle Variables can be written by device kernels")

    @property
    def funcname(self):
        return "".join(f.__name__ for f in self._kernels)

    @property
    def fieldset(self):
        return self._fieldset

    @property
    def pclass(self):
        return self._pclass

    def remove_deleted(self, pset):  # kernel.py:98-106
        """Remove all particles that signalled deletion."""
        indices = np.where(pset._data["state"] == StatusCode.Delete)[0]
        if len(indices) > 0:
            pset.remove_indices(indices)

    def merge(self, kernel):  # kernel.py:161-172
        if not isinstance(kernel, type(self)):
            raise TypeError(f"Cannot merge {type(kernel)} with {type(self)}. Both should be of type {type(self)}.")
        assert self.fieldset == kernel.fieldset, "Cannot merge kernels with different fieldsets"
        assert self.pclass == kernel.pclass, "Cannot merge kernels with different particle types"
        return type(self)(self._kernels + kernel._kernels, types.SimpleNamespace(fieldset=self.fieldset, _pclass=self.pclass))

    def check_fieldsets_in_kernels(self, kernel):
        """kernel.py:122-159, including its context side effects (RK45 defaults; the tolerance is divided by
        deg2m on a spherical mesh on EVERY Kernel construction, as in the reference)."""
        fs = self.fieldset
        if kernel is _k.AdvectionRK45:
            if "next_dt" not in [v.name for v in self.pclass.variables]:
                raise ValueError('ParticleClass requires a "next_dt" for AdvectionRK45 Kernel.')
            if not hasattr(fs, "RK45_tol"):
                warnings.warn("Setting RK45 tolerance to 10 m. Use fieldset.add_context('RK45_tol', [distance]) to change.",
                              KernelWarning, stacklevel=2)
                fs.add_context("RK45_tol", 10)
            if fs.U.grid._mesh.is_spherical():
                fs.context["RK45_tol"] = fs.RK45_tol / fs.U.grid.deg2m
            if not hasattr(fs, "RK45_min_dt"):
                warnings.warn("Setting RK45 minimum timestep to 1 s. Use fieldset.add_context('RK45_min_dt', [timestep]) to change.",
                              KernelWarning, stacklevel=2)
                fs.add_context("RK45_min_dt", 1)
            if not hasattr(fs, "RK45_max_dt"):
                warnings.warn("Setting RK45 maximum timestep to 1 day. Use fieldset.add_context('RK45_max_dt', [timestep]) to change.",
                              KernelWarning, stacklevel=2)
                fs.add_context("RK45_max_dt", 60 * 60 * 24)
        if kernel in (_k.AdvectionDiffusionM1, _k.AdvectionDiffusionEM, _k.DiffusionUniformKh):
            for name in ("Kh_zonal", "Kh_meridional"):
                if name not in fs.fields:
                    raise ValueError(f"{kernel.__name__} needs the field {name}")
            if kernel is not _k.DiffusionUniformKh and not hasattr(fs, "dres"):
                raise ValueError(f"{kernel.__name__} needs fieldset.add_context('dres', ...)")
        if kernel in (_k.AdvectionRK4_3D, _k.AdvectionRK2_3D) and "UVW" not in fs.fields:
            raise ValueError(f"{kernel.__name__} needs a W field (UVW)")

    def _have_guess0(self, data) -> int:
        g0 = self.fieldset.gridset[0]
        if g0.is_curvilinear and "X" in g0.axes:  # np.any(xi) over the guesses (index_search.py:269)
            xdim = max(g0.xdim, 1)
            return int(np.any(np.mod(data["ei"][:, 0].astype(np.int64), xdim) != 0))
        return 0

    def _try_jit(self, pset):
        """Compile the Python functions of the list into the device program (parcels_amd/jit.py).  On success the list is a pure
        device list: user ids PK_KERNEL_USER0 + j, the Variables the functions touch bound as device columns."""
        from . import jit

        self._jit_tried = True
        if not jit.jit_enabled():
            self.jit_report = "PARCELS_AMD_JIT=0"
            return
        try:
            ids, prog, dev_vars = jit.compile_kernel_list(self._kernels, _k.kernel_id, self._pclass, self._fieldset, pset._engine(),
                                                          samples=self.samples, device_variables=self.device_variables)
        except jit.NotTranslatable as e:
            self.jit_report = str(e)
            return
        except (OSError, RuntimeError) as e:  # no hipcc on this machine, a failed compilation: the host path still runs the list
            self.jit_report = f"{type(e).__name__}: {e}"
            warnings.warn(f"user kernels could not be compiled for the device ({self.jit_report.splitlines()[0]}); running them on the host path",
                          KernelWarning, stacklevel=3)
            return
        except Exception as e:  # noqa: BLE001 -- a construct the translator trips over must not take the run down: the host path runs any kernel
            self.jit_report = f"translator error {type(e).__name__}: {e}"
            warnings.warn(f"user kernels were not compiled for the device ({self.jit_report}); running them on the host path -- please report this kernel",
                          KernelWarning, stacklevel=3)
            return
        self.user_program = prog
        self.kernel_ids = ids
        self.device_variables = dev_vars
        self.host_functions = []
        self.jit_report = f"compiled {[f.__name__ for f in self._kernels if _k.kernel_id(f) is None]} into {prog.path}"

    def launch(self, pset, endtime, dt, have_guess0=0):
        """Device part of Kernel.execute: advance the BOUND, device-resident particle columns to ``endtime``.
        No host<->device copies; returns the engine statistics (steps, state histogram, kernel time)."""
        if self.host_functions:
            return self._launch_hosted(pset, endtime, dt)
        engine = pset._engine()
        engine.set_user_program(self.user_program)
        data = pset._data
        if "RK45_tol" in self.fieldset.context and "next_dt" not in data:
            # kernel.py:118-120: `particles.dt = particles.next_dt` runs whenever the fieldset has RK45_tol
            raise KeyError("next_dt: fieldset.context has RK45_tol (RK45 mode) but the ParticleClass has no next_dt Variable")
        sign = 1 if dt > 0 else -1
        t_start = pset._t_live if getattr(pset, "_t_live", None) is not None else float(np.nanmin(data["t"]) if sign > 0 else np.nanmax(data["t"]))
        every = getattr(pset, "resort_every", None)
        if every is None:
            every = getattr(type(pset), "RESORT_EVERY_DEFAULT", None)
        stats = engine.execute(self.kernel_ids, endtime=endtime, dt0=dt, context=self.fieldset.context, seed=pset.seed,
                               have_guess0=have_guess0, sort_by_cell=int(pset.sort_by_cell), t_start=t_start, samples=self.samples,
                               resort_every=every or None, in_place_variables=self.user_program is not None)
        pset._last_stats = stats
        return stats

    def _launch_hosted(self, pset, endtime, dt):
        """A kernel list with Python functions: columns to the host, the reference's loop there (hostkernels.execute_hosted: deletes
        and raises like kernel.py:233-245), columns back to the device for whatever follows (output snapshots, the next interval)."""
        from .hostkernels import execute_hosted

        engine = pset._engine()
        if getattr(engine, "_bound", None) is pset._data and len(pset) > 0:  # device-resident columns of the running execute()
            engine.d2h()
        try:
            stats = execute_hosted(self, pset, endtime, dt)
        finally:
            if len(pset) > 0:
                engine.device_variables = list(self.device_variables)
                engine.bind_particles(pset._data)
                engine.h2d()
        pset._last_stats = stats
        return stats

    @staticmethod
    def needs_host_pass(stats) -> bool:
        """Deletions must be compacted and error codes raised on the host (kernel.py:233-245)."""
        sc = stats["state_counts"]
        return any(code in sc for code in (StatusCode.Delete, StatusCode.StopAllExecution, *[c for c in sc if c >= StatusCode.Error]))

    @staticmethod
    def only_deletions(stats) -> bool:
        """The launch left particles in state Delete and nothing to raise: the compaction can stay on the device."""
        sc = stats["state_counts"]
        return StatusCode.Delete in sc and StatusCode.StopAllExecution not in sc and not any(c >= StatusCode.Error for c in sc)

    def finish_on_host(self, pset, first_code=None):
        """kernel.py:233-245 after the columns are back on the host: compact deleted particles, raise error codes (``first_code``: the
        first code to raise over ALL shards of a collective run, so that every rank raises the same exception type)."""
        data = pset._data
        deleted = data["state"] == StatusCode.Delete
        if np.any(deleted):
            pset.remove_indices(np.where(deleted)[0])
        raise_particle_errors(pset._data, first_code=first_code)

    def execute(self, pset, endtime, dt):
        """Advance every particle to ``endtime`` on the device (kernel.py:174-247), host columns in, host columns out."""
        if len(pset) == 0:
            return StatusCode.Success
        engine = pset._engine()
        pset._t_live = None
        if self.host_functions and not self._jit_tried:
            self._try_jit(pset)
        engine.device_variables = list(self.device_variables)
        engine.bind_particles(pset._data)
        engine.h2d()
        self.launch(pset, endtime, dt, have_guess0=self._have_guess0(pset._data))
        engine.d2h()
        self.finish_on_host(pset)
        return pset
