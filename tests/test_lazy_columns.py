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
        self.dev = {k: np.array(data.raw(k)) for k in data.raw_keys()}
        data._engine = self

    def launch(self):  # "kernels" move x and advance t
        self.dev["x"] += 1.0
        self.dev["t"] += 10.0
        self._bound.mark_launched(["x", "t", "state"])

    def d2h(self, cols=None):
        cols = list(self._bound.raw_keys()) if cols is None else cols
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


def test_c_fast_paths_go_through_the_bookkeeping():
    """dict(data), {**data} and f(**data) skipped __getitem__ of the dict SUBCLASS and saw the arrays of before the launch."""
    d, e = _cols(), _FakeEngine()
    e.bind_and_upload(d)
    e.launch()
    assert not isinstance(d, dict)
    a = dict(d)
    assert a["x"][0] == 1.0 and a["t"][0] == 10.0 and not d._stale and d._dirty == {"x", "t", "state", "age"}
    del a
    e.upload_dirty()
    e.launch()
    b = {**d}
    assert b["x"][0] == 2.0
    del b
    e.upload_dirty()
    e.launch()
    assert (lambda **kw: kw["x"][0])(**d) == 3.0


def test_a_held_array_is_refreshed_by_the_launch_and_its_writes_are_uploaded():
    """The reference hands out the live array (particleset.py:155-164): a reference taken BEFORE a launch sees the launch, and a write through
    it lands on the device with the next upload.  Columns nobody holds stay lazy."""
    d, e = _cols(), _FakeEngine()
    x = d["x"]  # held from before the set ever went to the device
    tail = d["t"][3:]  # a view holds its base
    e.bind_and_upload(d)
    e.launch()
    assert e.down == ["t", "x"] and x[0] == 1.0 and tail[0] == 10.0 and d._stale == {"state"} and d._dirty == {"x", "t"}
    x[:] = 50.0  # announced to nobody
    e.upload_dirty()
    assert e.up == ["t", "x"]
    e.launch()
    assert x[0] == 51.0 and d["x"] is x
    del x, tail
    e.upload_dirty()
    e.launch()  # nobody holds anything now: nothing comes down
    assert e.down == ["t", "x", "t", "x"] and d._stale == {"x", "t", "state"}


def test_clear_setdefault_popitem_and_ior_release_the_device_rows():
    for op in (lambda d: d.clear(), lambda d: d.setdefault("new", np.zeros(5)), lambda d: d.popitem(), lambda d: d.__ior__({"age": np.zeros(5)})):
        d, e = _cols(), _FakeEngine()
        e.bind_and_upload(d)
        e.launch()
        op(d)
        assert not d.resident() and not d._stale and sorted(e.down) == ["state", "t", "x"]
