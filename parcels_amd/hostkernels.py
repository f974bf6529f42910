"""Arbitrary Python kernels next to the device kernels (the reference's plug-in point #1: `def kernel(particles, fieldset)`,
src/parcels/_core/kernel.py:67-70; run by the loop of kernel.py:190-245).

A Python function cannot run inside a HIP kernel.  When a kernel list holds one, the loop of ``Kernel.execute`` runs HERE, on the
host columns, restated line by line -- and every built-in kernel of the list still runs on the GPU: its body is launched once per
iteration over the same ``evaluate_particles`` selection (``pk_exec_params.body_only`` + ``pk_particles_set_mask``), field sampling
inside a Python kernel (``fieldset.UV[particles]``) goes through ``pk_eval``.  The particle columns cross PCIe around every device
segment, so this is the slow path by construction: it exists so that ageing / beaching / custom-delete kernels written for the
reference run unchanged; pure built-in lists never come here.

``HostParticles`` is what the user function receives as ``particles``: attribute access to the columns of the selected particles
with NumPy semantics (``particles.age += particles.dt``, ``particles.dx[mask] -= 1``, ``particles[particles.t >= 4].state = ...``),
writing through to the particle set (particlesetview.py of the reference is the contract; this is an independent implementation on
NumPy's operator mix-in).
"""

from __future__ import annotations

import warnings

import numpy as np

from .statuscodes import StatusCode

__all__ = ["HostParticles", "execute_hosted"]


class _Column(np.lib.mixins.NDArrayOperatorsMixin):
    """One Variable of the selected particles: reads gather from the parent column, every mutation (``col += v``,
    ``col[sub] = v``, ``col[sub] += v``) scatters back into it in the column's own dtype (particlesetview.py:202-205)."""

    __slots__ = ("_parent", "_rows")

    def __init__(self, parent, rows):
        self._parent, self._rows = parent, rows

    def __array__(self, dtype=None, copy=None):
        a = self._parent[self._rows]
        return a.astype(dtype) if dtype is not None else a

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        vals = [np.asarray(x) if isinstance(x, _Column) else x for x in inputs]
        if out is not None:  # in-place: the result lands in the parent column
            targets = out if isinstance(out, tuple) else (out,)
            res = getattr(ufunc, method)(*vals, **kwargs)
            for t in targets:
                if isinstance(t, _Column):
                    t._parent[t._rows] = res
                else:
                    t[...] = res
            return targets[0] if len(targets) == 1 else targets
        return getattr(ufunc, method)(*vals, **kwargs)

    def __getitem__(self, sub):
        return np.asarray(self)[sub]

    def __setitem__(self, sub, value):
        rows = self._rows[sub]
        self._parent[rows] = np.asarray(value) if isinstance(value, _Column) else value

    def __len__(self):
        return len(self._rows)

    def __iter__(self):
        return iter(np.asarray(self))

    def __repr__(self):
        return repr(np.asarray(self))

    @property
    def dtype(self):
        return self._parent.dtype

    @property
    def shape(self):
        return (len(self._rows),) + self._parent.shape[1:]

    @property
    def size(self):
        return int(np.prod(self.shape))

    def astype(self, dt):
        return np.asarray(self).astype(dt)


class HostParticles:
    """The ``particles`` argument of a Python kernel: a selection of rows of the particle set's columns."""

    def __init__(self, data: dict, rows, by_mask=False):
        object.__setattr__(self, "_data", data)
        object.__setattr__(self, "_rows", np.asarray(rows, dtype=np.int64))
        # len() in the reference is `len(self._index)` (particlesetview.py:83-84), and what a kernel receives is a BOOLEAN mask over the whole
        # set -- as is every selection made from it, by mask or by integer indices (:44-74: `new_index[sel] = True`): len(particles) and
        # len(particles[anything]) are the size of the whole set there.  (len(particles.x) is the number of rows.)
        object.__setattr__(self, "_by_mask", bool(by_mask))

    def __getattr__(self, name):
        data = object.__getattribute__(self, "_data")
        if name not in data:
            raise AttributeError(f"particles have no Variable {name!r}")
        return _Column(data[name], object.__getattribute__(self, "_rows"))

    def __setattr__(self, name, value):
        if name not in self._data:
            raise AttributeError(f"particles have no Variable {name!r}")
        col = self._data[name]
        col[self._rows] = np.asarray(value) if isinstance(value, _Column) else value  # cast to the storage dtype

    def __getitem__(self, sub):
        """A sub-selection: boolean mask over these particles, integer index / array / slice into them, or np.where's tuple."""
        if isinstance(sub, tuple) and len(sub) == 1:
            sub = sub[0]
        if isinstance(sub, _Column):
            sub = np.asarray(sub)
        sub = np.asarray(sub) if isinstance(sub, (list, np.ndarray)) else sub
        if isinstance(sub, np.ndarray) and sub.dtype == bool and sub.shape[0] != len(self._rows):
            if sub.shape[0] == len(self._data["particle_id"]):  # a mask over the WHOLE set selects from it, whatever this selection was
                return HostParticles(self._data, np.flatnonzero(sub), by_mask=self._by_mask)  # (particlesetview.py:48-52: `new_index = arr`)
            raise IndexError(f"boolean index of length {sub.shape[0]} for a selection of {len(self._rows)} particles")
        rows = self._rows[sub]
        return HostParticles(self._data, np.atleast_1d(rows), by_mask=self._by_mask)

    def __len__(self):
        return len(self._data["particle_id"]) if self._by_mask else len(self._rows)

    def __repr__(self):
        return f"HostParticles({len(self._rows)} of {len(self._data['particle_id'])} particles)"


def _apply_sample_states(particles, st):
    """Field sampling inside a kernel marks the particles it fails on (field.py:307-378: out of bounds, through the surface,
    outside the time interval, NaN): the higher code wins, like the device kernels' status-code state machine."""
    if particles is None or st is None or not isinstance(particles, HostParticles):
        return
    cur = np.asarray(particles.state)
    st = np.asarray(st, dtype=cur.dtype)
    err = st >= int(StatusCode.Error)
    time_err = st == int(StatusCode.ErrorOutsideTimeInterval)
    new = np.where(time_err, st, np.where(err & (st > cur), st, cur))
    if np.any(new != cur):
        particles.state = new


def execute_hosted(kernel, pset, endtime, dt):
    """Kernel.execute (kernel.py:174-247) for a kernel list that contains Python functions.  Host columns in, host columns out
    (the caller re-uploads); returns the statistics dict of a launch."""
    from . import _hip, kernels as _k
    import ctypes as C

    engine = None  # created when a built-in kernel of the list needs the device (a list of Python functions alone never does)
    fs = kernel.fieldset
    sign = 1 if dt > 0 else -1
    rk45_mode = "RK45_tol" in fs.context
    d = pset._data
    d["state"][:] = int(StatusCode.Evaluate)  # :188
    steps = 0
    body_launches = 0
    first_body = True
    ev_states = (int(StatusCode.Evaluate), int(StatusCode.Repeat))

    def device_segment(ids, samples, mask):
        nonlocal body_launches, first_body, engine
        if engine is None:
            engine = pset._engine()
        data = pset._data
        engine.device_variables = list(kernel.device_variables)
        engine.bind_particles(data)
        engine.h2d()
        m = np.ascontiguousarray(mask, dtype=np.int32)
        engine.ctx.check(engine.lib.pk_particles_set_mask(engine.ctx.handle, m.ctypes.data_as(C.c_void_p)), "pk_particles_set_mask")
        prm = engine.make_params(ids, endtime=endtime, dt0=dt, context=fs.context, seed=pset.seed, reset_state=int(first_body),
                                 have_guess0=(kernel._have_guess0(data) if first_body else 1), sort_by_cell=0, samples=samples)
        prm.body_only = 1
        st = _hip.ExecStats()
        engine.ctx.check(engine.lib.pk_execute(engine.ctx.handle, C.byref(prm), C.byref(st)), "pk_execute (body)")
        engine.d2h()
        first_body = False
        body_launches += 1

    # the kernel list as runs of device kernels and single Python functions
    segments = []
    for slot, f in enumerate(kernel._kernels):
        kid = _k.kernel_id(f)
        if kid is None:
            segments.append(("py", f))
        elif segments and segments[-1][0] == "dev":
            segments[-1][1].append(kid)
            segments[-1][2].append(slot)
        else:
            segments.append(("dev", [kid], [slot]))

    while len(pset) > 0 and np.any(np.isin(pset._data["state"], ev_states)):  # :190
        d = pset._data
        tte = sign * (endtime - d["t"])
        ev = np.isin(d["state"], (int(StatusCode.Success), int(StatusCode.Evaluate))) & (tte >= 0)  # :193-195
        if not ev.any():
            break
        if sign == 1:  # :199-203 (every particle, also the ones not evaluated: the reference does the same)
            d["dt"][:] = np.maximum(np.minimum(d["dt"], tte), 0)
        else:
            d["dt"][:] = np.minimum(np.maximum(d["dt"], -tte), 0)
        rows = np.flatnonzero(ev)
        for seg in segments:  # :206-216
            with warnings.catch_warnings():
                from .field import FieldEvalWarning

                warnings.simplefilter("ignore", FieldEvalWarning)
                if seg[0] == "dev":
                    samples = {k: kernel.samples[s] for k, s in enumerate(seg[2]) if s in kernel.samples}
                    device_segment(seg[1], samples, ev)  # the device runs its own Repeat loop per kernel
                    d = pset._data
                else:
                    seg[1](HostParticles(d, rows, by_mask=True), fs)
                    rep = d["state"] == int(StatusCode.Repeat)
                    while rep.any():
                        seg[1](HostParticles(d, np.flatnonzero(rep), by_mask=True), fs)
                        rep = d["state"] == int(StatusCode.Repeat)
        upd = ev & np.isin(d["state"], (int(StatusCode.Evaluate), int(StatusCode.Success)))  # :219-222
        if upd.any():  # _position_update (:108-120), storage-dtype arithmetic of the columns
            for pos, delta in (("x", "dx"), ("y", "dy"), ("z", "dz")):
                d[pos][upd] += d[delta][upd]
            d["t"][upd] += d["dt"][upd]
            d["dx"][upd] = 0
            d["dy"][upd] = 0
            d["dz"][upd] = 0
            if rk45_mode:
                d["dt"][upd] = d["next_dt"][upd]
            steps += int(upd.sum())
        if not rk45_mode:
            d["dt"][:] = dt  # :225-226
        d["state"][(d["state"] == int(StatusCode.Evaluate)) & (d["t"] == endtime)] = int(StatusCode.EndofLoop)  # :229-230
        gone = d["state"] == int(StatusCode.Delete)  # :233
        if gone.any():
            pset.remove_indices(np.flatnonzero(gone))
            d = pset._data
        if np.any(d["state"] == int(StatusCode.StopAllExecution)):  # :236-237
            break
        from .engine import raise_particle_errors

        raise_particle_errors(d)  # :239-245
    d = pset._data
    codes, counts = np.unique(d["state"], return_counts=True)
    return {"steps": steps, "attempts": 0, "kernel_ms": 0.0, "sort_ms": 0.0, "launches": body_launches, "program": -1, "hosted": True,
            "first_error_iter": 0, "reran": 0, "state_counts": {int(c): int(n) for c, n in zip(codes, counts)}}
