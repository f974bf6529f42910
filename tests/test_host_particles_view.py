"""CPU: the `particles` object a Python kernel receives (parcels_amd/hostkernels.py: HostParticles) -- the semantics the reference's
ParticleSetView gives kernel authors (particlesetview.py): NumPy expressions on the selected particles, every mutation written
through to the particle set in the column's storage dtype."""

import numpy as np
import pytest

from parcels_amd.hostkernels import HostParticles, _apply_sample_states
from parcels_amd.statuscodes import StatusCode


def _data(n=6):
    return {"x": np.arange(n, dtype=np.float32), "dx": np.zeros(n, np.float32), "t": np.arange(n, dtype=np.float64), "dt": np.full(n, 0.5),
            "state": np.full(n, int(StatusCode.Evaluate), np.int32), "age": np.zeros(n, np.float32), "particle_id": np.arange(n), "ei": np.zeros((n, 1), np.int32)}


def test_attribute_reads_and_writes_go_through_the_selection():
    d = _data()
    p = HostParticles(d, [1, 3, 4])
    assert len(p) == 3 and np.array_equal(np.asarray(p.x), [1, 3, 4])
    p.age += p.dt  # float32 column += float64 column: stored as float32
    assert d["age"].dtype == np.float32 and np.array_equal(d["age"], [0, 0.5, 0, 0.5, 0.5, 0])
    p.dx += 0.1
    assert np.allclose(d["dx"], [0, 0.1, 0, 0.1, 0.1, 0])
    p.x = 7  # scalar broadcast
    assert np.array_equal(d["x"], [0, 7, 2, 7, 7, 5])
    p.state = np.where(np.asarray(p.t) >= 3, StatusCode.Delete, p.state)
    assert list(d["state"]) == [10, 10, 10, 30, 30, 10]
    with pytest.raises(AttributeError):
        p.nonexistent = 1
    with pytest.raises(AttributeError):
        _ = p.nonexistent


def test_sub_selections_and_item_assignment():
    d = _data()
    p = HostParticles(d, [0, 2, 3, 5])
    sub = p[p.t >= 3]  # boolean mask over the selection
    assert len(sub) == 2 and len(sub.x) == 2  # (a selection handed to a kernel by the loop answers len() like the reference's: by_mask=True)
    k = HostParticles(d, [0, 2, 3, 5], by_mask=True)
    assert len(k) == len(k[k.t >= 3]) == len(k[np.array([0, 1])]) == 6 and len(k[k.t >= 3].x) == 2
    sub.state = StatusCode.StopExecution
    assert list(d["state"]) == [10, 10, 10, 40, 10, 40]
    inds = np.where(p.state == StatusCode.StopExecution)  # np.where's tuple, as kernels write it
    p[inds].dx -= 1.0
    assert np.array_equal(d["dx"], [0, 0, 0, -1, 0, -1])
    p.dx[np.asarray(p.x) < 1] += 2.5  # item assignment on a column writes through too
    assert d["dx"][0] == 2.5 and d["dx"][2] == 0
    p[0].x = 9
    assert d["x"][0] == 9
    assert np.array_equal(np.asarray(p[1:3].x), [2, 3])
    assert (p.x + p.dx).dtype == np.float32 and np.argwhere(p.t >= 100).size == 0
    with pytest.raises(IndexError):
        p[np.ones(3, bool)]


def test_sampling_marks_failing_particles_like_the_state_machine():
    d = _data(4)
    d["state"][:] = [10, 60, 10, 0]
    p = HostParticles(d, [0, 1, 2, 3])
    _apply_sample_states(p, np.array([10, 51, 61, 70], np.int32))
    assert list(d["state"]) == [10, 60, 61, 70]  # the higher error code wins; OutsideTimeInterval is assigned (field.py:303-304)
