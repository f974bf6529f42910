"""LazyColumns: the SoA dict of a ParticleSet whose DEVICE copy may be newer than the NumPy arrays.

The reference re-reads every particle column from host memory on every ``ParticleSet.execute`` (particleset.py:355-470 builds a
fresh view per step); the first rounds of this engine uploaded all columns at the start of every call and downloaded all of them at
its end -- 35 ms of PCIe around a 9 ms launch on BASELINE config 2, although nothing on the host looks at the columns between two
calls of the usual script loop.  Here the columns stay in HBM from one call to the next and the host arrays are a lazy mirror:

* after a launch the device columns the kernels write are marked *stale* on the host (nothing is copied);
* reading a column (``data["x"]``, ``pset.x``, ``.items()``, ...) first downloads the stale column(s) touched
  (``pk_particles_d2h_columns``) and -- because the array that is handed out may be written in place, which a dict cannot see --
  marks them *dirty*;
* the next ``execute`` uploads only the dirty columns, into the device rows they belong to (``pk_particles_h2d_columns`` goes through
  the row order of the cell sort, which is kept), and nothing at all when the host did not touch the set;
* replacing a column's array (``data[k] = new``: remove_indices, add, ...) or binding another ParticleSet to the engine makes the host
  arrays authoritative again (everything stale is downloaded first).

``peek(k)`` reads a column with the promise not to write to it (no dirty mark).  References to an array obtained BEFORE a launch are
not refreshed by the launch, and a write through such an old reference is not seen; ask the dict again.  For that reason the mirror is lazy
only from ``ParticleSet.RESIDENT_MIN`` (1e5) particles on -- where the copies cost tens of milliseconds per call; smaller sets keep the eager
protocol (every call uploads and downloads everything: well under a millisecond), i.e. exactly the reference's aliasing behaviour.
"""

from __future__ import annotations

__all__ = ["LazyColumns", "raw_items", "readonly"]


class LazyColumns(dict):
    def __init__(self, data=()):
        super().__init__(data)
        self._engine = None  # the DeviceEngine whose device rows hold these columns (None: the host arrays are all there is)
        self._stale: set = set()  # columns whose host array is older than the device rows
        self._dirty: set = set()  # columns handed out since they last equalled the device rows

    # ---- engine-facing -------------------------------------------------------------------------------------------
    def resident(self) -> bool:
        e = self._engine
        return e is not None and getattr(e, "_bound", None) is self

    def raw(self, k):
        """The host array as it is (engine internals: pointers, dtypes, shapes)."""
        return dict.__getitem__(self, k)

    def set_raw(self, k, v):
        dict.__setitem__(self, k, v)

    def mark_launched(self, names):
        """A launch wrote these device columns: their host arrays are stale (and whatever the host wrote before was uploaded)."""
        names = set(names) & set(self.keys())
        self._stale |= names
        self._dirty -= names

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
        """Read-only access: current values, no dirty mark."""
        if k in self._stale:
            self.sync([k])
        return dict.__getitem__(self, k)

    # ---- dict protocol -------------------------------------------------------------------------------------------
    def __getitem__(self, k):
        if k in self._stale:
            self.sync([k])
        if self._engine is not None:
            self._dirty.add(k)
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def _touch_all(self):
        if self._stale:
            self.sync()
        if self._engine is not None:
            self._dirty |= set(self.keys())

    def items(self):
        self._touch_all()
        return dict.items(self)

    def values(self):
        self._touch_all()
        return dict.values(self)

    def copy(self):
        self._touch_all()
        return dict(dict.items(self))

    def __setitem__(self, k, v):
        # a new array for a column: the bound pointers no longer describe this set
        self.release()
        dict.__setitem__(self, k, v)

    def __delitem__(self, k):
        self.release()
        dict.__delitem__(self, k)

    def pop(self, k, *default):
        self.release()
        return dict.pop(self, k, *default)

    def update(self, *a, **kw):
        self.release()
        dict.update(self, *a, **kw)

    def __reduce__(self):  # pickling / copy.deepcopy: a plain snapshot of current values
        self._touch_all()
        return (LazyColumns, (dict(dict.items(self)),))


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
        return iter(dict.keys(self._lc))

    def __len__(self):
        return len(self._lc)

    def keys(self):
        return dict.keys(self._lc)

    def get(self, k, default=None):
        return self._lc.peek(k) if k in self._lc else default

    def items(self):
        return [(k, self._lc.peek(k)) for k in dict.keys(self._lc)]


def readonly(data):
    return _ReadOnly(data) if isinstance(data, LazyColumns) else data


def raw_items(data):
    """(name, array as it is) pairs: no download, no dirty mark (the caller knows which columns are current)."""
    if isinstance(data, LazyColumns):
        return [(k, data.raw(k)) for k in dict.keys(data)]
    return list(data.items())
