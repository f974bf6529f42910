"""The known-answer interpolation tests the reference holds for this path (tests/test_interpolation.py:33-205): XLinear, XNearest,
XLinearInvdistLandTracer, XFreeslip / XPartialslip and the mesh-type unit conversion on its 3 x 4 x 4 x 4 ramp field.  The same
expected values are demanded of the CPU oracle (po_eval; runs everywhere) and of the HIP path (pk_eval through Field.eval /
VectorField.eval; `-m gpu`)."""

import numpy as np
import pytest

TZYX = ("time", "depth", "YG", "XG")


def ramp_case(**kw):
    """The `field` fixture of the reference: +1 per x, +2 per y, +3 per z, +10 per t; t = 0, 2, 4 s; z, y, x = 0..3; Z padding HIGH."""
    z0 = np.array([[0.0, 1.0, 2.0, 3.0], [2.0, 3.0, 4.0, 5.0], [4.0, 5.0, 6.0, 7.0], [6.0, 7.0, 8.0, 9.0]])
    spatial = np.array([z0, z0 + 3, z0 + 6, z0 + 9])
    data = np.array([spatial, spatial + 10, spatial + 20])
    case = dict(name="ramp", kind="sample", mesh="flat", lon=np.arange(4.0), lat=np.arange(4.0), depth=np.arange(4.0), x_pad="low", y_pad="low",
                z_pad="high", time_s=np.array([0.0, 2.0, 4.0]), fields={"P": data}, field_dims={"P": TZYX}, cgrid=False,
                scalar_interp={"P": "XLinear"}, sample_field="P", kernels=[], spatial_dtype="float64", dt=1.0, runtime=None, seed=0)
    case.update(kw)
    return case


def points(case, t, z, y, x):
    t, z, y, x = np.broadcast_arrays(*(np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (t, z, y, x)))
    case.update(t0=t.copy(), z=z.copy(), y=y.copy(), x=x.copy())
    return case


RAW = [  # test_interpolation.py:77-118
    pytest.param("XLinear", [0, 1], [0, 0], [0.49, 0.49], [0.51, 0.51], [1.49, 6.49], id="Linear-1"),
    pytest.param("XLinear", 1, 2.5, 0.49, 0.51, 13.99, id="Linear-2"),
    pytest.param("XLinear", [0, 1, 1], [0, 0, 2.5], [0.49, 0.49, 0.49], [0.51, 0.51, 0.51], [1.49, 6.49, 13.99], id="Linear-3"),
    pytest.param("XLinearInvdistLandTracer", 1, 2.5, 0.49, 0.51, 13.99, id="LinearInvDistLand"),
    pytest.param("XNearest", [0, 3], [0.2, 0.2], [0.2, 0.2], [0.51, 0.51], [1.0, 16.0], id="Nearest"),
]
INVDIST = [  # test_interpolation.py:157-186: data 1 with a 2 x 2 block of land (0) in the middle
    pytest.param(1, 0, 0.5, 0.5, 1.0, id="ocean-corner"),
    pytest.param(1, 0, 1.5, 1.5, 0.0, id="all-land"),
    pytest.param([0, 1], [0, 2], [0.5, 0.5], [0.5, 0.5], 1.0, id="two-times"),
    pytest.param([0, 1], [0, 2], [0.5, 1.5], [0.5, 1.5], [1.0, 0.0], id="mixed"),
]
SLIP = [  # test_interpolation.py:121-154
    ("partial", 1, 0, 0, 0.0, [[1.0], [1.0]]),
    ("free", 1, 0, 0.5, 1.5, [[1.0], [0.5]]),
    ("partial", 1, 0, 2.5, 1.5, [[0.75], [0.5]]),
    ("free", 1, 0, 2.5, 1.5, [[1.0], [0.5]]),
    ("partial", 1, 0, 1.5, 0.5, [[0.5], [0.75]]),
    ("free", 1, 0, 1.5, 0.5, [[0.5], [1.0]]),
    ("free", [1, 0], [0, 2], [1.5, 1.5], [2.5, 0.5], [[0.5, 0.5], [1.0, 1.0]]),
]


def land_block(case, names=("P",)):
    for n in names:
        d = np.ones_like(case["fields"]["P"] if "P" in case["fields"] else next(iter(case["fields"].values())))
        d[:, :, 1:3, 1:3] = 0.0
        case["fields"][n] = d
    return case


def slip_case(slip, mesh, t, z, y, x):
    case = ramp_case(kind="advect", mesh=mesh, slip=slip)
    data = np.ones((3, 4, 4, 4))
    data[:, :, 1:3, 1:3] = 0.0
    case["fields"] = {"U": data, "V": data.copy()}
    case["field_dims"] = {"U": TZYX, "V": TZYX}
    return points(case, t, z, y, x)


def slip_expected(expected, mesh, y):
    e = np.array(expected, dtype=np.float64)
    if mesh == "spherical":
        e[0] = e[0] / (1852 * 60.0 * np.cos(np.radians(np.atleast_1d(np.asarray(y, dtype=np.float64)))))
        e[1] = e[1] / (1852 * 60.0)
    return e


# ---- the CPU oracle --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("interp, t, z, y, x, expected", RAW)
def test_oracle_raw_interpolation(interp, t, z, y, x, expected):
    from oracle import c_oracle as co

    case = points(ramp_case(scalar_interp={"P": interp}), t, z, y, x)
    np.testing.assert_allclose(co.sample_case(case)["value"], np.atleast_1d(expected), rtol=0, atol=1e-14)


@pytest.mark.parametrize("t, z, y, x, expected", INVDIST)
def test_oracle_invdistland_interpolation(t, z, y, x, expected):
    from oracle import c_oracle as co

    case = points(land_block(ramp_case(scalar_interp={"P": "XLinearInvdistLandTracer"})), t, z, y, x)
    np.testing.assert_array_almost_equal(co.sample_case(case)["value"], np.broadcast_to(expected, case["x"].shape))


# ---- the HIP path ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("interp, t, z, y, x, expected", RAW)
def test_raw_2d_interpolation(gpu, interp, t, z, y, x, expected):
    from case_utils import sample_hip

    case = points(ramp_case(scalar_interp={"P": interp}), t, z, y, x)
    np.testing.assert_allclose(sample_hip(case)["value"], np.atleast_1d(expected), rtol=0, atol=1e-14)


@pytest.mark.gpu
@pytest.mark.parametrize("t, z, y, x, expected", INVDIST)
def test_invdistland_interpolation(gpu, t, z, y, x, expected):
    from case_utils import sample_hip

    case = points(land_block(ramp_case(scalar_interp={"P": "XLinearInvdistLandTracer"})), t, z, y, x)
    np.testing.assert_array_almost_equal(sample_hip(case)["value"], np.broadcast_to(expected, case["x"].shape))


@pytest.mark.gpu
@pytest.mark.parametrize("mesh", ["flat", "spherical"])
@pytest.mark.parametrize("slip, t, z, y, x, expected", SLIP)
def test_spatial_slip_interpolation(gpu, slip, t, z, y, x, expected, mesh):
    from case_utils import build_fieldset

    case = slip_case(slip, mesh, t, z, y, x)
    fs = build_fieldset(case)
    u, v = fs.UV.eval(case["t0"], case["z"], case["y"], case["x"])
    np.testing.assert_array_almost_equal(np.array([u, v]), slip_expected(expected, mesh, y))


@pytest.mark.gpu
@pytest.mark.parametrize("mesh", ["spherical", "flat"])
def test_interpolation_mesh_type(gpu, mesh):  # test_interpolation.py:189-205
    import parcels_amd as pa
    from test_gpu_semantics import simple_uv_dataset

    fs = pa.FieldSet.from_sgrid_conventions(simple_uv_dataset(mesh=mesh, u=1.0), mesh=mesh)
    lat = 30.0
    u_expected = 1.0 if mesh == "flat" else 1.0 / (1852 * 60 * np.cos(np.radians(lat)))
    assert fs.U.eval(0.0, 0, lat, 0) == 1.0  # a velocity component on its own is not converted
    assert fs.V.eval(0.0, 0, lat, 0) == 0.0
    u, v = fs.UV.eval(0.0, 0, lat, 0)
    assert np.isclose(u, u_expected, atol=1e-7)
    assert v == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("field_name, location, expected", [  # tests/test_field.py:187-214
    ("U", (0.0, 0.0, 0.0, 5e6), 0.0),
    ("UV", (0.0, 0.0, 0.0, 5e6), [[0.0], [0.0]]),
    ("U", (0.0, 0.0, 5e6, 0.0), 0.0),
    ("UV", (0.0, 0.0, 5e6, 0.0), [[0.0], [0.0]]),
    ("U", (0.0, 5e6, 0.0, 0.0), 0.0),
    ("UV", (0.0, 5e6, 0.0, 0.0), [[0.0], [0.0]]),
])
def test_field_eval_out_of_bounds_structured(gpu, field_name, location, expected):
    """Field.eval outside the grid returns 0 with the reference's FieldEvalWarning."""
    import parcels_amd as pa
    from test_gpu_semantics import simple_uv_dataset

    fs = pa.FieldSet.from_sgrid_conventions(simple_uv_dataset(mesh="flat", u=1.0, v=2.0), mesh="flat")
    field = getattr(fs, field_name)
    with pytest.warns(pa.FieldEvalWarning, match="Some interpolated values are out-of-bounds. These values are set to 0. Treat carefully."):
        np.testing.assert_allclose(field.eval(*location), expected)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("error", pa.FieldEvalWarning)
        inside = field.eval(0.0, 0.0, 0.0, 0.0)  # no warning inside the grid
    np.testing.assert_allclose(inside, 1.0 if field_name == "U" else [[1.0], [2.0]])
