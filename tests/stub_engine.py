"""TEST INFRASTRUCTURE: an engine for host-path tests on machines without a GPU.  parcels_amd's host path (hostkernels.execute_hosted: the
reference's loop on the host columns) asks the device for exactly two things when a Python kernel samples a field -- the values at explicit
points with the sample's status codes (DeviceEngine.sample -> pk_eval) and the cell of a point (DeviceEngine.search -> pk_search).  Here the
CPU oracle answers both (po_eval, po_search_1d / po_hash_query), so that everything AROUND the sample -- which particles are marked, how `ei`
is updated, what the loop does with the states -- can be compared with the reference's real loop in the CPU suite.  Only tests use this."""
import ctypes as C
import types

import numpy as np


class OracleBackedEngine:
    def __init__(self, fieldset, case):
        from oracle import c_oracle as co

        self.co, self.mc, self.case = co, co.MarshalledCase(case), case
        self.grids = fieldset.gridset
        self.device, self.windowed, self.last_sample_state = 0, False, None
        self.nslots_request = None
        self.points_f32 = 0
        self.ctx = types.SimpleNamespace(check=lambda rc, what=None: None, handle=None)
        self.lib = types.SimpleNamespace(pk_set_option=self._set_option)
        self.samples = 0
        self.nonfinite_points = False  # some sample point was NaN / inf (the reference's answer then depends on the rest of the batch)

    def _set_option(self, handle, name, value):
        if name == b"eval_points_f32":
            self.points_f32 = int(value)
        return 0

    def sample(self, name, t, z, y, x):
        assert not self.points_f32, "the oracle's po_eval takes float64 points (use a float64 particle class in these tests)"
        co, mc = self.co, self.mc
        t, z, y, x = np.broadcast_arrays(*(np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (t, z, y, x)))
        t, z, y, x = (np.ascontiguousarray(v) for v in (t, z, y, x))
        m = x.shape[0]
        self.nonfinite_points |= not all(np.all(np.isfinite(a)) for a in (t, z, y, x))
        u, v, w, st = np.zeros(m), np.zeros(m), np.zeros(m), np.zeros(m, np.int32)
        what = {"UV": -1, "UVW": -2}.get(name, mc.field_index.get(name))
        prm = mc.params(kernels=[], endtime=0.0, dt0=1.0)
        rc = co.lib().po_eval(mc.grids, mc.fields, C.byref(prm), C.c_int32(what), C.c_int64(m), co._ptr(t), co._ptr(z), co._ptr(y), co._ptr(x),
                              co._ptr(u), co._ptr(v), co._ptr(w), co._ptr(st))
        assert rc == 0
        self.samples += 1
        self.last_sample_state = st
        return u, v, w

    def search(self, igrid, z, y, x):
        assert igrid == 0
        n = len(np.atleast_1d(x))
        data = {"x": np.asarray(x, dtype=np.float64), "y": np.asarray(y, dtype=np.float64), "z": np.asarray(z, dtype=np.float64), "ei": np.zeros((n, 1), np.int32)}
        self.co.populate_indices(self.mc, data)
        return data["ei"][:, 0]


def install(fieldset, case):
    """Give `fieldset` an oracle-backed engine (FieldSet._engine_or_create returns the engine it already has)."""
    eng = OracleBackedEngine(fieldset, case)
    fieldset.__dict__["_engine"] = eng
    return eng
