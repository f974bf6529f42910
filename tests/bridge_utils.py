"""Stand-ins with the attribute surface of the reference's FieldSet / XGrid / Field / VectorField / ParticleSet objects (what
parcels_amd.reference_bridge reads), built from an oracle case dict.  The GPU box has no reference tree: the GPU tests drive the bridge
with these; tests/test_reference_bridge.py checks on the CPU -- where the reference is present -- that the reference's REAL classes give
the bridge the same FieldSet as these stand-ins do."""
from types import SimpleNamespace as NS

import numpy as np


def _named(name):
    return type(name, (), {})()


def standin_fieldset(case):
    pad = lambda s: NS(value=s)  # noqa: E731  (the reference's Padding enum: .value is the lower-case name)
    depth = case.get("depth")
    md = NS(node_dimensions=("XG", "YG"), node_coordinates=("lon", "lat"),
            face_dimensions=(NS(face="XC", node="XG", padding=pad(case.get("x_pad", "low"))), NS(face="YC", node="YG", padding=pad(case.get("y_pad", "low")))),
            vertical_dimensions=(NS(face="ZC", node="depth", padding=pad(case.get("z_pad", "both"))),) if depth is not None else None)
    mesh = lambda m: NS(is_spherical=lambda: m == "spherical")  # noqa: E731
    grid = NS(sgrid_metadata=md, lon=np.asarray(case["lon"]), lat=np.asarray(case["lat"]), depth=None if depth is None else np.asarray(depth),
              _mesh=mesh(case["mesh"]))
    ts = case.get("time_s")
    tcoord = None
    if ts is not None and len(ts) > 1:
        tcoord = NS(data=(np.asarray(ts, dtype=float) * 1e9).round().astype("int64").astype("timedelta64[ns]"))
    fields = {}
    for name, arr in case["fields"].items():
        dims = tuple(case["field_dims"][name])
        da = NS(data=np.asarray(arr), dims=dims)
        if tcoord is not None and "time" in dims:
            da.time = tcoord
        fields[name] = NS(name=name, data=da, grid=grid, interp_method=_named("XLinear"))
    vname = "CGrid_Velocity" if case.get("cgrid") or any(d in ("XC", "YC") for dd in case["field_dims"].values() for d in dd) else "XLinear_Velocity"
    if case.get("slip"):
        vname = {"free": "XFreeslip", "partial": "XPartialslip"}[case["slip"]]
    if "U" in fields and "V" in fields:
        fields["UV"] = NS(name="UV", U=fields["U"], V=fields["V"], W=None, interp_method=_named(vname))
        if "W" in fields:
            fields["UVW"] = NS(name="UVW", U=fields["U"], V=fields["V"], W=fields["W"], interp_method=_named(vname))
    gridset = [grid]
    if case.get("constants"):
        cg = NS(sgrid_metadata=NS(node_dimensions=("XG", "YG"), node_coordinates=("lon", "lat"),
                                  face_dimensions=(NS(face="XC", node="XG", padding=pad("low")), NS(face="YC", node="YG", padding=pad("low"))),
                                  vertical_dimensions=None),
                lon=np.zeros(1), lat=np.zeros(1), depth=None, _mesh=mesh(case.get("const_mesh", "flat")))
        for name, val in case["constants"].items():
            fields[name] = NS(name=name, data=NS(data=np.full((1, 1, 1, 1), val), dims=("mockT", "mockZ", "YG", "XG")), grid=cg,
                              interp_method=_named("XConstantField"))
        gridset.append(cg)
    return NS(fields=fields, gridset=gridset, context=dict(case.get("context") or {}))


def assert_same_fieldset(a, b):
    """Two parcels_amd.FieldSet objects describe the same device FieldSet: grids, fields, interpolators, time axes, context."""
    assert list(a.fields) == list(b.fields), (list(a.fields), list(b.fields))
    assert len(a.gridset) == len(b.gridset)
    for ga, gb in zip(a.gridset, b.gridset):
        assert ga.axes == gb.axes and ga._mesh.is_spherical() == gb._mesh.is_spherical()
        assert [ga.get_axis_dim(ax) for ax in ga.axes] == [gb.get_axis_dim(ax) for ax in gb.axes] and ga.offsets() == gb.offsets()
        for ax in ("lon", "lat") + (("depth",) if "Z" in ga.axes else ()):
            xa, xb = np.asarray(getattr(ga, ax)), np.asarray(getattr(gb, ax))
            assert xa.dtype == xb.dtype and np.array_equal(xa, xb), ax
    for name in a.fields:
        fa, fb = a.fields[name], b.fields[name]
        assert type(fa) is type(fb) and type(fa.interp_method) is type(fb.interp_method), name
        assert a.gridset.index(fa.grid) == b.gridset.index(fb.grid)
        if hasattr(fa, "U"):
            assert [c.name for c in (fa.U, fa.V, fa.W) if c is not None] == [c.name for c in (fb.U, fb.V, fb.W) if c is not None]
            continue
        da, db = fa.data, fb.data
        assert tuple(da.dims) == tuple(db.dims) and np.asarray(da.data).dtype == np.asarray(db.data).dtype, (name, da.dims, db.dims)
        assert np.array_equal(np.asarray(da.data), np.asarray(db.data)), name
        ta, tb = fa.model.time_flt, fb.model.time_flt
        assert (ta is None) == (tb is None) and (ta is None or np.array_equal(ta, tb)), name
        assert (fa.time_interval is None) == (fb.time_interval is None)
    assert dict(a.context) == dict(b.context)
