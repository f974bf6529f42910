"""LazyColumns: the SoA mapping of a ParticleSet whose DEVICE copy may be newer than the NumPy arrays.

The reference re-reads every particle column from host memory on every ``ParticleSet.execute`` (particleset.py:355-470 builds a
fresh view per step); the first rounds of this engine uploaded all columns at the start of every call and downloaded all of them at
its end -- 35 ms of PCIe around a 9 ms launch on BASELINE config 2, although nothing on the host looks at the columns between two
calls of the usual script loop.  Here the columns stay in HBM from one call to the next and the host arrays are a lazy mirror:

* every column has ONE persistent ndarray (the memory the library copies to and from); the mapping hands out that very object, like the
  reference hands out the live array (particleset.py:155-164);
* after a launch the device columns the kernels write are marked *stale* on the host -- EXCEPT those whose ndarray somebody outside the set
  still holds (``sys.getrefcount``: a variable, a view, a container): such a column is downloaded right away, so the holder sees the new
  values in place exactly as with the reference, and marked *dirty*, because the holder may write through it at any time;
* reading a column (``data["x"]``, ``pset.x``, ``.items()``, ``dict(data)``, ``**data`` ...) first downloads it if it is stale
  (``pk_particles_d2h_columns``) and marks it dirty -- the array that is handed out may be written in place, which no mapping can see;
* the next ``execute`` uploads only the dirty columns, into the device rows they belong to (``pk_particles_h2d_columns`` goes through
  the row order of the cell sort, which is kept), and nothing at all when the host did not touch the set and holds none of its arrays;
* replacing a column's array (``data[k] = new``: remove_indices, add, ...) or binding another ParticleSet to the engine makes the host
  arrays authoritative again (everything stale is downloaded first).

Rounds 5's version subclassed ``dict``: the C fast paths of ``dict(data)``, ``{**data}``, ``np.savez(p, **data)`` skip ``__getitem__`` and
handed out arrays of before the launch, and an array obtained before a launch was neither refreshed by it nor uploaded when
written -- which is why the mirror was lazy only from 1e5 particles on.  This one is a ``collections.abc.MutableMapping`` over a private dict
(every access goes through the sync / dirty bookkeeping) and follows held references, so the aliasing is the reference's at every size.

``peek(k)`` reads a column with the promise not to write to it and not to keep it (no dirty mark).
"""

from __future__ import annotations

import sys
from collections.abc import MutableMapping

__all__ = ["LazyColumns", "raw_items", "readonly"]

# references to a column's ndarray that are the set's own when `_held` looks at it: the private dict's, the local variable's and the
# argument of sys.getrefcount itself
_OWN_REFS = 3


class LazyColumns(MutableMapping):
    def __init__(self, data=()):
        self._cols = dict(data)
        self._engine = None  # the DeviceEngine whose device rows hold these columns (None: the host arrays are all there is)
        self._stale: set = set()  # columns whose host array is older than the device rows
        self._dirty: set = set()  # columns handed out (or still held outside) since they last equalled the device rows
        self._extra_refs: dict = {}  # references the ENGINE keeps to a column's array (DeviceEngine: the float32 next_dt column and its shadow)

    # ---- engine-facing -------------------------------------------------------------------------------------------
    def resident(self) -> bool:
        e = self._engine
        return e is not None and getattr(e, "_bound", None) is self

    def raw(self, k):
        """The host array as it is (engine internals: pointers, dtypes, shapes)."""
        return self._cols[k]

    def set_raw(self, k, v):
        self._cols[k] = v

    def raw_keys(self):
        return self._cols.keys()

    def _held(self, k) -> bool:
        """Somebody outside the set references this column's ndarray (directly, or through a view whose `.base` it is)."""
        a = self._cols[k]
        return sys.getrefcount(a) > _OWN_REFS + self._extra_refs.get(k, 0)

    def held(self, names=None):
        return [k for k in (self._cols if names is None else names) if k in self._cols and self._held(k)]

    def mark_launched(self, names):
        """A launch wrote these device columns: their host arrays are stale (and whatever the host wrote before was uploaded).  Columns
        whose array is held outside the set are refreshed now and stay dirty (module docstring); returns their names."""
        names = set(names) & set(self._cols)
        self._stale |= names
        self._dirty -= names
        held = sorted(k for k in names if self._held(k))
        if held and self.resident():
            self._engine.d2h(held)  # (DeviceEngine.d2h clears them from _stale)
            self._dirty |= set(held)
        return held

    def sync(self, names=None):
        """Download the stale columns (all of them, or those of `names`)."""
        need = set(self._stale) if names is None else (self._stale & set(names))
        if not need:
            return
        if not self.resident():
            raise RuntimeError("LazyColumns: stale columns but the device rows are gone (internal error)")
        self._engine.d2h(sorted(need))  # (DeviceEngine.d2h clears them from _stale)

    def release(self):
        """Make the host arrays authoritative: download what is stale, forget the device rows."""
        if self._engine is not None:
            if self._stale:
                self.sync()
            self._engine = None
        self._stale.clear()
        self._dirty.clear()

    def peek(self, k):
        """Read-only access: current values, no dirty mark (the caller neither writes to the array nor keeps it)."""
        if k in self._stale:
            self.sync([k])
        return self._cols[k]

    # ---- mapping protocol ----------------------------------------------------------------------------------------
    def __getitem__(self, k):
        if k in self._stale:
            self.sync([k])
        a = self._cols[k]
        if self._engine is not None:
            self._dirty.add(k)
        return a

    def __iter__(self):
        return iter(self._cols)

    def __len__(self):
        return len(self._cols)

    def __contains__(self, k):
        return k in self._cols

    def keys(self):
        return self._cols.keys()

    def _touch_all(self):
        if self._stale:
            self.sync()
        if self._engine is not None:
            self._dirty |= set(self._cols)

    def items(self):
        self._touch_all()
        return self._cols.items()

    def values(self):
        self._touch_all()
        return self._cols.values()

    def copy(self):
        self._touch_all()
        return dict(self._cols)

    def __setitem__(self, k, v):
        # a new array for a column: the bound pointers no longer describe this set
        self.release()
        self._cols[k] = v

    def __delitem__(self, k):
        self.release()
        del self._cols[k]

    def clear(self):
        self.release()
        self._cols.clear()

    def update(self, *a, **kw):
        self.release()
        self._cols.update(*a, **kw)

    def __ior__(self, other):  # (a dict has `|=`; so does its stand-in)
        self.update(other)
        return self

    def __repr__(self):
        return f"LazyColumns({list(self._cols)}, stale={sorted(self._stale)}, dirty={sorted(self._dirty)}, resident={self.resident()})"

    def __reduce__(self):  # pickling / copy.deepcopy: a plain snapshot of current values
        self._touch_all()
        return (LazyColumns, (dict(self._cols),))


class _ReadOnly:
    """Mapping view of a LazyColumns set for code that only READS columns (write-out filters, error reports): item access downloads what
    is stale but leaves no dirty mark, so the next execute uploads nothing because of it."""

    def __init__(self, lc):
        self._lc = lc

    def __getitem__(self, k):
        return self._lc.peek(k)

    def __contains__(self, k):
        return k in self._lc

    def __iter__(self):
        return iter(self._lc.raw_keys())

    def __len__(self):
        return len(self._lc)

    def keys(self):
        return self._lc.raw_keys()

    def get(self, k, default=None):
        return self._lc.peek(k) if k in self._lc else default

    def items(self):
        return [(k, self._lc.peek(k)) for k in self._lc.raw_keys()]


def readonly(data):
    return _ReadOnly(data) if isinstance(data, LazyColumns) else data


def raw_items(data):
    """(name, array as it is) pairs: no download, no dirty mark (the caller knows which columns are current)."""
    if isinstance(data, LazyColumns):
        return [(k, data.raw(k)) for k in data.raw_keys()]
    return list(data.items())
