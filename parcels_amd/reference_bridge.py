"""The binding a Parcels maintainer would add to route ``Kernel.execute`` (src/parcels/_core/kernel.py:174-247) through
``libparcels_hip.so`` -- as running code rather than a sketch.

It works on the REFERENCE's own objects by attribute (nothing of the reference is imported here, so the module loads -- and is
tested -- with or without the reference installed):

* ``fieldset_from_reference(fieldset)``: the reference's ``FieldSet`` (``fields`` / ``gridset`` / ``context``; its ``XGrid`` s with
  ``sgrid_metadata``, ``lon`` / ``lat`` / ``depth``, ``_mesh``; its ``Field`` s with ``data`` (time, z, y, x), ``grid``,
  ``interp_method``; its ``VectorField`` s with ``U`` / ``V`` / ``W``) as a ``parcels_amd.FieldSet`` over the same NumPy buffers (no copy
  of C-contiguous field data), which ``to_device()`` turns into grid / field descriptors of include/parcels_hip.h;
* ``HipBackend(fieldset)``: owns the device copy; ``execute(pset, kernel_functions, endtime, dt)`` binds the reference's SoA dict
  ``pset._data`` (particle.py:182-222 -- same column names and dtypes as ``pk_particles_desc``), runs the fused launch with the
  reference's stop-at-first-error semantics and copies the columns back;
* ``install(kernel_module)``: wraps ``Kernel.execute`` so that kernel lists made of built-ins (matched by function name: AdvectionRK4,
  AdvectionRK4_3D, AdvectionRK45, AdvectionDiffusionM1, ...) and of elementwise user-written kernels (compiled into the launch by
  parcels_amd/jit.py) go to the GPU and everything else -- any other Python kernel, an unstructured grid, an interpolator without a
  device form -- falls through to the untouched NumPy loop.

tests/test_reference_bridge.py runs ``fieldset_from_reference`` and the installed dispatch against the reference's real ``XGrid`` /
``Field`` / ``VectorField`` / ``ParticleSet`` / ``Kernel`` classes (CPU, when /root/reference is present) and the same backend end to
end on the GPU against the reference-generated fixtures (attribute-compatible stand-ins: the reference does not travel to the GPU box).
"""

from __future__ import annotations

import numpy as np

from . import kernels as _k
from .dataset import Dataset
from .field import StructuredModelData
from .fieldset import FieldSet
from .sgrid import FaceNodePadding, Padding, SGrid2DMetadata
from .statuscodes import StatusCode

__all__ = ["fieldset_from_reference", "HipBackend", "install", "UnsupportedByDevice"]


class UnsupportedByDevice(Exception):
    """The FieldSet / kernel list has no device form: the caller keeps the reference's NumPy path."""


def _padding(p) -> Padding:
    return Padding(str(getattr(p, "value", p)).lower())


def _metadata(md) -> SGrid2DMetadata:
    """SGrid2DMetadata of the reference (_sgrid/core.py:70-190) -> ours, field by field."""
    faces = tuple(FaceNodePadding(f.face, f.node, _padding(f.padding)) for f in md.face_dimensions)
    vert = md.vertical_dimensions
    vert = tuple(FaceNodePadding(f.face, f.node, _padding(f.padding)) for f in vert) if vert else None
    return SGrid2DMetadata(node_dimensions=tuple(md.node_dimensions), node_coordinates=tuple(md.node_coordinates or ("lon", "lat")),
                           face_dimensions=faces, vertical_dimensions=vert)


def _interpolator(method):
    """The device form of a reference interpolator object, by class name (interpolators/_xinterpolators.py)."""
    import parcels_amd as pa

    name = type(method).__name__
    cls = getattr(pa, name, None)
    if cls is None or not isinstance(cls, type):
        raise UnsupportedByDevice(f"interpolator {name} has no device form")
    return cls()


def _is_vector(f) -> bool:
    return hasattr(f, "U") and hasattr(f, "V")


def _time_coord(da):
    t = getattr(da, "time", None)
    if t is None:
        coords = getattr(da, "coords", None)
        t = coords["time"] if coords is not None and "time" in coords else None
    return None if t is None else np.asarray(getattr(t, "data", t))


def fieldset_from_reference(ref_fs) -> FieldSet:
    """A ``parcels_amd.FieldSet`` over the data of a reference FieldSet (see the module docstring)."""
    scalars = {n: f for n, f in ref_fs.fields.items() if not _is_vector(f)}
    vectors = {n: f for n, f in ref_fs.fields.items() if _is_vector(f)}
    grids = []
    for f in scalars.values():
        if not any(f.grid is g for g in grids):
            grids.append(f.grid)
    models, constants = [], []
    for g in grids:
        if not hasattr(g, "sgrid_metadata"):
            raise UnsupportedByDevice(f"{type(g).__name__} is not a structured grid")
        on_grid = {n: f for n, f in scalars.items() if f.grid is g}
        mesh = "spherical" if g._mesh.is_spherical() else "flat"
        if all(type(f.interp_method).__name__ == "XConstantField" for f in on_grid.values()):
            for n, f in on_grid.items():  # constant fields live on their own 1 x 1 grid (model.py:292-317)
                constants.append((n, float(np.asarray(f.data.data).reshape(-1)[0]), mesh))
            continue
        md = _metadata(g.sgrid_metadata)
        xn, yn = md.node_dimensions
        lon, lat = np.asarray(g.lon), np.asarray(g.lat)
        coords = {"lon": ((xn,), lon), "lat": ((yn,), lat)} if lon.ndim == 1 else {"lon": ((yn, xn), lon), "lat": ((yn, xn), lat)}
        grid_dims = {xn, yn} | {f.face for f in md.face_dimensions}
        if md.vertical_dimensions:
            zn = md.vertical_dimensions[0].node
            coords["depth"] = ((zn,), np.asarray(g.depth))
            grid_dims |= {zn, md.vertical_dimensions[0].face}
        data_vars = {}
        for n, f in on_grid.items():
            da = f.data
            a = np.asarray(da.data)
            dims = tuple(da.dims)
            tv = _time_coord(da) if "time" in dims else None
            if tv is not None and tv.size > 1 and "time" not in coords:
                coords["time"] = (("time",), tv)
            keep = [i for i, d in enumerate(dims) if d in grid_dims or (d == "time" and a.shape[i] > 1)]
            data_vars[n] = (tuple(dims[i] for i in keep), a.reshape([a.shape[i] for i in keep]))
        vec = {vn: tuple(c.name for c in (v.U, v.V, getattr(v, "W", None)) if c is not None)
               for vn, v in vectors.items() if v.U.name in on_grid}
        model = StructuredModelData(Dataset(data_vars, coords, sgrid=md), mesh, vec)
        model._fields = model.construct_fields()
        models.append(model)
    if not models:
        raise UnsupportedByDevice("the FieldSet has no field on a structured grid")
    fs = FieldSet(models)
    for n, f in scalars.items():
        if n in fs.fields:
            fs.fields[n].interp_method = _interpolator(f.interp_method)
    for vn, v in vectors.items():
        if vn not in fs.fields:
            raise UnsupportedByDevice(f"vector field {vn} spans grids")
        fs.fields[vn].interp_method = _interpolator(v.interp_method)
    for n, val, mesh in constants:
        fs.add_constant_field(n, val, mesh=mesh)
    for k, v in dict(ref_fs.context).items():
        fs.add_context(k, v)
    return fs


class HipBackend:
    """Device copy of one reference FieldSet + the launch of its built-in kernel lists."""

    def __init__(self, ref_fieldset, device: int = 0, nslots=None, seed: int = 0):
        self.ref_fieldset = ref_fieldset
        self.fieldset = fieldset_from_reference(ref_fieldset)
        self.device, self.nslots, self.seed = int(device), nslots, int(seed)
        self._engine = None
        self.last_stats = None
        self._plans = {}
        self.jit_report = None

    @property
    def engine(self):
        if self._engine is None:  # grids, hash tables, field levels: created on first use (needs the GPU)
            self._engine = self.fieldset.to_device(device=self.device, nslots=self.nslots)._engine
        return self._engine

    @staticmethod
    def builtin_id(f):
        """PK_KERNEL_* id of a reference kernel function (matched by name), None for anything else."""
        mine = getattr(_k, getattr(f, "__name__", ""), None)
        return _k.kernel_id(mine) if mine is not None and getattr(mine, "_pk_sample", None) is None else None

    @classmethod
    def kernel_ids(cls, kernel_functions):
        """Ids of a list made of built-ins only, or None when it holds a function without a built-in device form."""
        ids = [cls.builtin_id(f) for f in kernel_functions]
        return None if any(i is None for i in ids) else ids

    def plan(self, kernel_functions, pclass=None):
        """(kernel ids, user program or None, device Variable names) for a kernel list, or None when the list stays on the reference's
        NumPy loop.  User-written functions are compiled into the launch when they are elementwise (parcels_amd/jit.py; needs the
        particle class for the Variables' dtypes)."""
        for k, v in dict(self.ref_fieldset.context).items():  # constants are compiled in: the plan is per context
            self.fieldset.context[k] = v
        key = (tuple(kernel_functions), repr(sorted(self.fieldset.context.items(), key=lambda kv: kv[0])))
        if key in self._plans:
            return self._plans[key]
        ids = self.kernel_ids(kernel_functions)
        plan = (ids, None, []) if ids is not None else None
        if ids is None and pclass is not None:
            from . import jit

            if jit.jit_enabled():
                try:
                    plan = jit.compile_kernel_list(list(kernel_functions), self.builtin_id, pclass, self.fieldset, self.engine)
                except jit.NotTranslatable as e:
                    self.jit_report = str(e)
                except Exception as e:  # a failing compilation must never break a run the NumPy loop can do
                    self.jit_report = f"{type(e).__name__}: {e}"
        self._plans[key] = plan
        return plan

    def supports(self, kernel_functions, pset=None) -> bool:
        plan = self.plan(kernel_functions, getattr(pset, "_pclass", None))
        if plan is None:
            return False
        names = {getattr(f, "__name__", "") for f in kernel_functions}
        ctx = self.ref_fieldset.context
        if "AdvectionRK45" in names and not all(k in ctx for k in ("RK45_tol", "RK45_min_dt", "RK45_max_dt")):
            return False  # (Kernel.__init__ of the reference sets them: kernel.py:122-159)
        if names & {"AdvectionRK4_3D", "AdvectionRK2_3D"} and "UVW" not in self.fieldset.fields:
            return False
        return True

    def _have_guess0(self, data) -> int:
        g0 = self.fieldset.gridset[0]
        if g0.is_curvilinear and "X" in g0.axes:  # np.any(xi) over the guesses (index_search.py:269)
            return int(np.any(np.mod(data["ei"][:, 0].astype(np.int64), max(g0.xdim, 1)) != 0))
        return 0

    def execute(self, pset, kernel_functions, endtime: float, dt: float) -> dict:
        """kernel.py:188-232 for the whole batch on the device: ``pset._data`` in, ``pset._data`` out (states included; deleting and
        raising -- kernel.py:233-245 -- stay with the caller, which has the reference's own code for both)."""
        plan = self.plan(kernel_functions, getattr(pset, "_pclass", None))
        if plan is None:
            raise UnsupportedByDevice("a kernel of the list has no device form")
        ids, program, dev_vars = plan
        data = pset._data
        n = len(data["t"])
        if n == 0:
            return {"steps": 0, "state_counts": {}}
        for k, v in dict(self.ref_fieldset.context).items():  # RK45 defaults arrive with Kernel.__init__, after the backend was built
            self.fieldset.context[k] = v
        eng = self.engine
        eng.device_variables = list(dev_vars)
        eng.set_user_program(program)
        eng.bind_particles(data)
        eng.h2d()
        sign = 1 if dt > 0 else -1
        t_start = float(np.nanmin(data["t"]) if sign > 0 else np.nanmax(data["t"]))
        st = eng.execute(ids, endtime=float(endtime), dt0=float(dt), context=self.fieldset.context, seed=self.seed,
                         have_guess0=self._have_guess0(data), sort_by_cell=0, t_start=t_start, in_place_variables=program is not None)
        eng.d2h()
        self.last_stats = st
        return st


def install(kernel_module, device: int = 0, min_particles: int = 0):
    """Wrap ``kernel_module.Kernel.execute`` (the reference's parcels._core.kernel).  Returns a function that undoes it.

    The wrapper keeps one HipBackend per FieldSet (on the FieldSet object), hands a call to the GPU when every kernel is a built-in
    with a device form and the FieldSet is structured, then runs the reference's own tail -- ``remove_deleted`` and the ErrorsToThrow
    loop -- on the columns that came back; any other call goes to the original method untouched."""
    Kernel = kernel_module.Kernel
    original = Kernel.execute
    errors_to_throw = getattr(kernel_module, "ErrorsToThrow", {})
    stop_all = int(StatusCode.StopAllExecution)

    def execute(self, pset, endtime, dt):
        fs = self._fieldset
        backend = getattr(fs, "_hip_backend", None)
        try:
            if backend is None:
                backend = HipBackend(fs, device=device)
                object.__setattr__(fs, "_hip_backend", backend)
            ok = backend is not False and len(pset) >= min_particles and backend.supports(self._kernels, pset)
        except UnsupportedByDevice:
            object.__setattr__(fs, "_hip_backend", False)
            ok = False
        if not ok:
            return original(self, pset, endtime, dt)
        if len(pset) == 0:
            return pset
        backend.execute(pset, self._kernels, endtime, dt)
        self.remove_deleted(pset)  # kernel.py:233
        state = pset._data["state"]
        if np.any(state == stop_all):  # :236-237
            return type(state.flat[0])(stop_all) if state.size else stop_all
        for code, func in errors_to_throw.items():  # :239-245
            inds = state == code
            if np.any(inds):
                d = pset._data
                if int(code) == int(StatusCode.ErrorOutsideTimeInterval):
                    func(d["t"][inds])
                else:
                    func(d["z"][inds], d["y"][inds], d["x"][inds])
        return pset

    Kernel.execute = execute

    def uninstall():
        Kernel.execute = original

    return uninstall
