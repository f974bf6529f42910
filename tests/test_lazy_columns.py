"""parcels_amd/columns.py: the lazy host mirror of device-resident particle columns, against a stand-in engine (pure Python "device")."""

import numpy as np

from parcels_amd.columns import LazyColumns, raw_items, readonly


class _FakeEngine:
    """Device rows = a dict of arrays; d2h / h2d copy by column name and count what crossed."""

    def __init__(self):
        self._bound, self.dev, self.down, self.up = None, {}, [], []

    def bind_and_upload(self, data):
        prev = self._bound
        for lc in (prev, data):
            if isinstance(lc, LazyColumns):
                lc.release()
        self._bound = data
        self.dev = {k: np.array(data.raw(k)) for k in dict.keys(data)}
        data._engine = self

    def launch(self):  # "kernels" move x and advance t
        self.dev["x"] += 1.0
        self.dev["t"] += 10.0
        self._bound.mark_launched(["x", "t", "state"])

    def d2h(self, cols=None):
        cols = list(dict.keys(self._bound)) if cols is None else cols
        for k in cols:
            self._bound.raw(k)[...] = self.dev[k]
            self.down.append(k)
        self._bound._stale -= set(cols)

    def upload_dirty(self):
        for k in sorted(self._bound._dirty):
            self.dev[k][...] = self._bound.raw(k)
            self.up.append(k)
        self._bound._dirty.clear()


def _cols(n=5):
    return LazyColumns({"x": np.arange(n, dtype=float), "t": np.zeros(n), "state": np.zeros(n, np.int32), "age": np.ones(n)})


def test_plain_dict_behaviour_without_an_engine():
    d = _cols()
    assert not d.resident() and d["x"][2] == 2.0 and not d._dirty and not d._stale
    d["x"] = d["x"] * 2
    assert d["x"][2] == 4.0 and sorted(k for k, _ in d.items()) == ["age", "state", "t", "x"]


def test_reads_download_only_the_touched_stale_columns_and_mark_them_dirty():
    d, e = _cols(), _FakeEngine()
    e.bind_and_upload(d)
    e.launch()
    assert d._stale == {"x", "t", "state"} and e.down == []
    assert d.raw("x")[0] == 0.0  # (the host array itself is old)
    assert d["x"][0] == 1.0 and e.down == ["x"] and d._stale == {"t", "state"} and d._dirty == {"x"}
    assert d.peek("t")[0] == 10.0 and e.down == ["x", "t"] and "t" not in d._dirty
    assert readonly(d)["state"][0] == 0 and "state" not in d._dirty and not d._stale
    assert d["age"][0] == 1.0 and e.down == ["x", "t", "state"]  # a host-only column costs nothing (but is handed out: dirty)
    e.upload_dirty()
    assert e.up == ["age", "x"]


def test_untouched_set_moves_nothing_between_launches():
    d, e = _cols(), _FakeEngine()
    e.bind_and_upload(d)
    for _ in range(3):
        e.launch()
        e.upload_dirty()
    assert e.down == [] and e.up == [] and len(d.raw("x")) == 5
    assert d["x"][0] == 3.0 and d["t"][0] == 30.0


def test_host_write_between_launches_reaches_the_device():
    d, e = _cols(), _FakeEngine()
    e.bind_and_upload(d)
    e.launch()
    d["x"][:] = 100.0  # download, then an in-place write the dict cannot see: the column was marked dirty when it was handed out
    e.upload_dirty()
    e.launch()
    assert d["x"][0] == 101.0 and e.up == ["x"]


def test_replacing_an_array_or_binding_another_set_makes_the_host_authoritative():
    d, e = _cols(), _FakeEngine()
    e.bind_and_upload(d)
    e.launch()
    d["x"] = np.delete(d["x"], [0])  # remove_indices: everything stale comes down first, the set leaves the device
    assert not d.resident() and not d._stale and d.raw("t")[0] == 10.0 and sorted(e.down) == ["state", "t", "x"]
    d2 = _cols(3)
    e.bind_and_upload(d)
    e.launch()
    e.bind_and_upload(d2)  # another ParticleSet takes the engine: the first one gets its columns back
    assert not d.resident() and d.raw("t")[0] == 20.0 and d2.resident()


def test_items_values_copy_and_raw_items():
    d, e = _cols(), _FakeEngine()
    e.bind_and_upload(d)
    e.launch()
    assert [k for k, _ in raw_items(d)] == ["x", "t", "state", "age"] and e.down == []
    snap = {k: np.array(v) for k, v in d.items()}
    assert snap["x"][0] == 1.0 and not d._stale and d._dirty == {"x", "t", "state", "age"}
    import copy

    c = copy.deepcopy(d)
    assert isinstance(c, LazyColumns) and not c.resident() and c["t"][0] == 10.0
