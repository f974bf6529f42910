"""CPU: parcels_amd.XGrid against the reference's REAL XGrid (src/parcels/_core/xgrid.py + basegrid.py under oracle/ref_shim.py) for the
grids of the parity cases: axes, cell counts per axis, deg2m, the coordinate arrays, `ravel_index` on random in-range and out-of-bounds
index codes (the `ei` a kernel writes, wrapped to int32 like the particle column), and the field-dimension -> axis mapping."""
import numpy as np
import pytest

from oracle import ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")


def _cases():
    from oracle import cases

    return [cases.rect_agrid_case("g_sph", mesh="spherical", kernels=["AdvectionRK4"], seed=1, npart=4, nx=9, ny=7, nz=4, nt=3),
            cases.rect_agrid_case("g_flat", mesh="flat", kernels=["AdvectionRK4_3D"], seed=2, npart=4, with_w=True, nx=5, ny=6, nz=3),
            cases.rect_cgrid_case("g_cgrid", mesh="spherical", kernels=["AdvectionRK4_3D"], seed=3, npart=4),
            cases.curv_cgrid_case("g_curv", mesh="spherical", kernels=["AdvectionRK4_3D"], seed=4, npart=4, nx=12, ny=10)]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_grid_properties_and_ravel_index(case):
    from case_utils import build_fieldset
    from oracle.make_golden import build_ref_fieldset

    ref_fs, rg = build_ref_fieldset(case)
    mg = build_fieldset(case).gridset[0]
    assert list(rg.axes) == list(mg.axes)
    for ax in rg.axes:
        assert rg.get_axis_dim(ax) == mg.get_axis_dim(ax), ax
    assert (rg.xdim, rg.ydim) == (mg.xdim, mg.ydim) and (("Z" not in rg.axes) or rg.zdim == mg.zdim)
    assert float(rg.deg2m) == float(mg.deg2m) and rg._mesh.is_spherical() == mg._mesh.is_spherical()
    for name in ("lon", "lat") + (("depth",) if "Z" in rg.axes else ()):
        a, b = np.asarray(getattr(rg, name)), np.asarray(getattr(mg, name))
        assert a.dtype == b.dtype and np.array_equal(a, b), name
    rng = np.random.default_rng(5)
    n = 200
    idx = {}
    for ax in rg.axes:
        v = rng.integers(0, rg.get_axis_dim(ax), size=n)
        v[rng.random(n) < 0.1] = -1   # RIGHT_OUT_OF_BOUNDS
        v[rng.random(n) < 0.05] = -2  # LEFT_OUT_OF_BOUNDS
        v[rng.random(n) < 0.03] = -3  # GRID_SEARCH_ERROR
        idx[ax] = v
    full = {"X": idx.get("X", np.zeros(n, int)), "Y": idx.get("Y", np.zeros(n, int)), "Z": idx.get("Z", np.zeros(n, int))}
    ra = np.asarray(rg.ravel_index(full)).astype(np.int32)
    rb = np.asarray(mg.ravel_index(full)).astype(np.int32)
    assert np.array_equal(ra, rb)
    # ... and back (what a kernel does with particles.ei: tests/test_particlefile.py's Get_XiYi), error codes included
    ua, ub = rg.unravel_index(ra), mg.unravel_index(rb)
    assert list(ua) == list(ub)
    for ax in ua:
        assert np.array_equal(ua[ax], ub[ax]), ax
    # which axis of the grid each dimension of a field lies on
    for name, dims in case["field_dims"].items():
        real = [d for d in dims if not str(d).startswith("mock") and d != "time"]
        assert dict(rg.get_axis_dim_mapping(real)) == dict(mg.get_axis_dim_mapping(real)), name
