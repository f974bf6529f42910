"""TEST INFRASTRUCTURE ONLY -- deterministic parity cases (inputs) shared by the golden generator, the oracle
tests and the GPU parity tests.

A case is a plain dict in NumPy terms:

  mesh                "flat" | "spherical"
  lon, lat, depth     node coordinates in the dataset's own dtype (lon/lat 1-D rectilinear or 2-D curvilinear;
                      depth may be None)
  x_pad,y_pad,z_pad   SGRID padding of the face dims ("low" | "high" | "both" | "none")
  time_s              seconds of the time levels (None/len 1 => time-invariant fields)
  fields, field_dims  name -> TZYX ndarray, name -> 4 dimension names
                      (nodes XG/YG/depth, faces XC/YC/ZC, "time"; "mockT"/"mockZ"/... for absent axes)
  cgrid               UV/UVW interpolator is CGrid_Velocity instead of XLinear_Velocity
  constants           name -> value of constant fields (FieldSet.add_constant_field), const_mesh
  context             fieldset.context entries (RK45_tol, RK45_min_dt, RK45_max_dt, dres)
  kernels             list of built-in kernel names
  spatial_dtype       "float32" (default Particle) | "float64"
  x, y, z, t0         release positions / times (z None => default, t0 None => 0)
  dt, runtime|endtime execute() arguments (seconds)
  seed                RNG seed of the stochastic kernels

The synthetic datasets restate the reference's analytic generators
(src/parcels/_datasets/structured/generated.py:10-366) and the shapes its hot-path tests use.
"""

from __future__ import annotations

import math

import numpy as np

TZYX_NODE = ("time", "depth", "YG", "XG")


def _rng(seed):
    return np.random.default_rng(seed)


def smooth_random_field(rng, shape, scale=1.0, dtype=np.float64):
    """Random but smooth-ish field: white noise lightly box-filtered along y and x."""
    a = rng.standard_normal(shape)
    a = (a + np.roll(a, 1, axis=-1) + np.roll(a, -1, axis=-1)) / 3.0
    a = (a + np.roll(a, 1, axis=-2) + np.roll(a, -1, axis=-2)) / 3.0
    return (scale * a).astype(dtype)


def rect_agrid_case(
    name,
    *,
    mesh,
    kernels,
    seed=0,
    nx=36,
    ny=18,
    nz=5,
    nt=4,
    npart=300,
    field_dtype=np.float64,
    spatial_dtype="float64",
    coord_dtype=np.float64,
    with_w=False,
    dt=3600.0,
    runtime=None,
    vel=None,
    margin=0.12,
    stagger=False,
    level_dt=86400.0,
    wscale=0.004,
):
    rng = _rng(seed)
    if mesh == "spherical":
        lon = np.linspace(0, 360, nx).astype(coord_dtype)
        lat = np.linspace(-80, 80, ny).astype(coord_dtype)
        vel = 1.0 if vel is None else vel  # m/s
    else:
        lon = np.linspace(0, 4.0e5, nx).astype(coord_dtype)
        lat = np.linspace(0, 2.0e5, ny).astype(coord_dtype)
        vel = 0.4 if vel is None else vel
    depth = np.linspace(0, 5000, nz).astype(coord_dtype)
    time_s = np.arange(nt) * level_dt
    shp = (nt, nz, ny, nx)
    fields = {"U": smooth_random_field(rng, shp, vel, field_dtype), "V": smooth_random_field(rng, shp, vel, field_dtype)}
    dims = {"U": TZYX_NODE, "V": TZYX_NODE}
    if with_w:
        fields["W"] = smooth_random_field(rng, shp, wscale * vel, field_dtype)
        dims["W"] = TZYX_NODE
    lx, ly = float(lon[-1] - lon[0]), float(lat[-1] - lat[0])
    x = rng.uniform(lon[0] + margin * lx, lon[-1] - margin * lx, npart)
    y = rng.uniform(lat[0] + margin * ly, lat[-1] - margin * ly, npart)
    z = rng.uniform(200, 4800, npart)
    case = dict(
        name=name,
        mesh=mesh,
        lon=lon,
        lat=lat,
        depth=depth,
        x_pad="low",
        y_pad="low",
        z_pad="both",
        time_s=time_s,
        fields=fields,
        field_dims=dims,
        cgrid=False,
        kernels=list(kernels),
        spatial_dtype=spatial_dtype,
        x=x,
        y=y,
        z=z,
        t0=None,
        dt=dt,
        runtime=runtime if runtime is not None else 30 * abs(dt),
        seed=seed,
    )
    if stagger:
        case["t0"] = rng.integers(0, 6, npart) * abs(dt) * 0.5
    return case


# ---- restated analytic datasets of the reference (generated.py) ---------------------------------------------------


def peninsula_case(name, *, mesh="flat", grid_type="A", kernels=("AdvectionRK4",), xdim=100, ydim=50, npart=20,
                   spatial_dtype="float32", dt=1800.0, runtime=23 * 3600.0):
    """generated.py:206-298 (flow around an idealised peninsula; P is the conserved streamfunction)."""
    domainsizeX, domainsizeY = (1.0e5, 5.0e4)
    La = np.linspace(0, domainsizeX, xdim, dtype=np.float32)
    Wa = np.linspace(0, domainsizeY, ydim, dtype=np.float32)
    u0 = 1
    x0 = domainsizeX / 2
    R = 0.32 * domainsizeX / 2
    P = np.zeros((ydim, xdim), dtype=np.float32)
    U = np.zeros_like(P)
    V = np.zeros_like(P)
    x, y = np.meshgrid(La, Wa, sparse=True, indexing="xy")
    P[:, :] = u0 * R**2 * y / ((x - x0) ** 2 + y**2) - u0 * y
    landpoints = P >= 0.0
    P[landpoints] = 0.0
    if grid_type == "A":
        U[:, :] = u0 - u0 * R**2 * ((x - x0) ** 2 - y**2) / (((x - x0) ** 2 + y**2) ** 2)
        V[:, :] = -2 * u0 * R**2 * ((x - x0) * y) / (((x - x0) ** 2 + y**2) ** 2)
        U[landpoints] = 0.0
        V[landpoints] = 0.0
        udims = ("mockT", "mockZ", "YC", "XC")
        vdims = ("mockT", "mockZ", "YC", "XC")
    else:
        U = np.zeros(P.shape)
        V = np.zeros(P.shape)
        U[1:, :] = -(P[1:, :] - P[:-1, :]) / (Wa[1] - Wa[0])
        V[:, 1:] = (P[:, 1:] - P[:, :-1]) / (La[1] - La[0])
        udims = ("mockT", "mockZ", "YG", "XC")
        vdims = ("mockT", "mockZ", "YC", "XG")
    lon = La / 1852.0 / 60.0 if mesh == "spherical" else La
    lat = Wa / 1852.0 / 60.0 if mesh == "spherical" else Wa
    # release line of tests/test_advection.py:405-408
    if mesh == "spherical":
        px = np.full(npart, 3.0e3 / 1852.0 / 60.0)
        py = np.linspace(3.0e3, 47.0e3, npart) / 1852.0 / 60.0
    else:
        px = np.full(npart, 3.0e3)
        py = np.linspace(3.0e3, 47.0e3, npart)
    return dict(
        name=name,
        mesh=mesh,
        lon=lon,
        lat=lat,
        depth=None,
        x_pad="low",
        y_pad="low",
        z_pad="both",
        time_s=None,
        fields={"U": U[None, None], "V": V[None, None], "P": P[None, None]},
        field_dims={"U": udims, "V": vdims, "P": ("mockT", "mockZ", "YC", "XC")},
        cgrid=(grid_type == "C"),
        kernels=list(kernels),
        spatial_dtype=spatial_dtype,
        x=px,
        y=py,
        z=None,
        t0=None,
        dt=dt,
        runtime=runtime,
        seed=0,
    )


def stommel_case(name, *, grid_type="A", kernels=("AdvectionRK4",), xdim=60, ydim=60, npart=16,
                 spatial_dtype="float32", dt=3600.0, runtime=20 * 86400.0):
    """generated.py:301-366 (Stommel gyre; P conserved)."""
    a = b = 10000 * 1e3
    scalefac = 0.05
    dx, dy = a / xdim, b / ydim
    lon = np.linspace(0, a, xdim, dtype=np.float32)
    lat = np.linspace(0, b, ydim, dtype=np.float32)
    U = np.zeros((lat.size, lon.size), dtype=np.float32)
    V = np.zeros((lat.size, lon.size), dtype=np.float32)
    P = np.zeros((lat.size, lon.size), dtype=np.float32)
    beta = 2e-11
    r = 1 / (11.6 * 86400)
    es = r / (beta * a)
    for j in range(lat.size):
        for i in range(lon.size):
            xi = lon[i] / a
            yi = lat[j] / b
            P[j, i] = (1 - math.exp(-xi / es) - xi) * math.pi * np.sin(math.pi * yi) * scalefac
            if grid_type == "A":
                U[j, i] = -(1 - math.exp(-xi / es) - xi) * math.pi**2 * np.cos(math.pi * yi) * scalefac
                V[j, i] = (math.exp(-xi / es) / es - 1) * math.pi * np.sin(math.pi * yi) * scalefac
    if grid_type == "C":
        U[1:, :] = -(P[1:, :] - P[0:-1, :]) / dy * b
        V[:, 1:] = (P[:, 1:] - P[:, 0:-1]) / dx * a
        udims = ("mockT", "mockZ", "YG", "XC")
        vdims = ("mockT", "mockZ", "YC", "XG")
    else:
        udims = ("mockT", "mockZ", "YC", "XC")
        vdims = ("mockT", "mockZ", "YC", "XC")
    px = np.linspace(a * 0.1, a * 0.4, npart)
    py = np.full(npart, b * 0.5)
    return dict(
        name=name,
        mesh="flat",
        lon=lon,
        lat=lat,
        depth=None,
        x_pad="low",
        y_pad="low",
        z_pad="both",
        time_s=None,
        fields={"U": U[None, None], "V": V[None, None], "P": P[None, None]},
        field_dims={"U": udims, "V": vdims, "P": ("mockT", "mockZ", "YG", "XG")},
        cgrid=(grid_type == "C"),
        kernels=list(kernels),
        spatial_dtype=spatial_dtype,
        x=px,
        y=py,
        z=None,
        t0=None,
        dt=dt,
        runtime=runtime,
        seed=0,
    )


def moving_eddy_case(name, *, kernels=("AdvectionRK4",), spatial_dtype="float32", dt=900.0, runtime=6 * 3600.0,
                     context=None):
    """generated.py:94-140 (eddy moving in time, no spatial variation); closed form in tests/test_advection.py:254-307."""
    f, u_0, u_g = 1.0e-4, 0.3, 0.04
    xdim = ydim = 2
    lon = np.linspace(0, 25000, xdim, dtype=np.float32)
    lat = np.linspace(0, 25000, ydim, dtype=np.float32)
    time_s = np.arange(0, 7 * 3600, 60).astype(np.float64)
    U = np.zeros((len(time_s), 1, ydim, xdim), dtype=np.float32)
    V = np.zeros((len(time_s), 1, ydim, xdim), dtype=np.float32)
    for t in range(len(time_s)):
        U[t] = u_g + (u_0 - u_g) * np.cos(f * time_s[t])
        V[t] = -(u_0 - u_g) * np.sin(f * time_s[t])
    case = dict(
        name=name,
        mesh="flat",
        lon=lon,
        lat=lat,
        depth=np.array([0.0]),
        x_pad="low",
        y_pad="high",
        z_pad="both",
        time_s=time_s,
        fields={"U": U, "V": V},
        field_dims={"U": TZYX_NODE, "V": TZYX_NODE},
        cgrid=False,
        kernels=list(kernels),
        spatial_dtype=spatial_dtype,
        x=np.array([12000.0, 12500.0]),
        y=np.array([12500.0, 12000.0]),
        z=np.array([0.0, 0.0]),
        t0=None,
        dt=dt,
        runtime=runtime,
        seed=0,
        attrs=dict(f=f, u_0=u_0, u_g=u_g),
    )
    if context:
        case["context"] = dict(context)
    return case


def rect_cgrid_case(name, *, mesh, kernels, seed=0, nx=30, ny=20, nz=6, nt=3, npart=300, field_dtype=np.float32,
                    spatial_dtype="float64", with_w=True, dt=3600.0, runtime=None, vel=0.5):
    """NEMO-like rectilinear C-grid (X,Y LOW padding, Z HIGH: convert.py:382-398): U on x-faces, V on y-faces,
    W on z-faces, all with the same array extents as the node grid."""
    rng = _rng(seed)
    if mesh == "spherical":
        lon = np.linspace(-20, 40, nx)
        lat = np.linspace(-30, 30, ny)
    else:
        lon = np.linspace(0, 3.0e5, nx)
        lat = np.linspace(0, 2.0e5, ny)
    depth = np.linspace(0, 3000, nz)
    time_s = np.arange(nt) * 86400.0
    shp = (nt, nz, ny, nx)
    fields = {"U": smooth_random_field(rng, shp, vel, field_dtype), "V": smooth_random_field(rng, shp, vel, field_dtype)}
    dims = {"U": ("time", "ZC", "YC", "XG"), "V": ("time", "ZC", "YG", "XC")}
    if with_w:
        fields["W"] = smooth_random_field(rng, shp, 0.01 * vel, field_dtype)
        dims["W"] = ("time", "depth", "YC", "XC")
    lx, ly = lon[-1] - lon[0], lat[-1] - lat[0]
    x = rng.uniform(lon[0] + 0.15 * lx, lon[-1] - 0.15 * lx, npart)
    y = rng.uniform(lat[0] + 0.15 * ly, lat[-1] - 0.15 * ly, npart)
    z = rng.uniform(300, 2700, npart)
    return dict(
        name=name,
        mesh=mesh,
        lon=lon,
        lat=lat,
        depth=depth,
        x_pad="low",
        y_pad="low",
        z_pad="high",
        time_s=time_s,
        fields=fields,
        field_dims=dims,
        cgrid=True,
        kernels=list(kernels),
        spatial_dtype=spatial_dtype,
        x=x,
        y=y,
        z=z,
        t0=None,
        dt=dt,
        runtime=runtime if runtime is not None else 24 * abs(dt),
        seed=seed,
    )


def curvilinear_grid(nx, ny, *, mesh, seed=0):
    """Smoothly warped + rotated lon/lat mesh (in the spirit of _datasets/structured/generic.py:13-62)."""
    i = np.arange(nx)[None, :] / (nx - 1)
    j = np.arange(ny)[:, None] / (ny - 1)
    if mesh == "spherical":
        lon0 = -30 + 70 * i + 0 * j
        lat0 = 20 + 45 * j + 0 * i
        lon = lon0 + 4.0 * np.sin(2 * np.pi * j) * (0.3 + i) + 6.0 * j
        lat = lat0 + 3.0 * np.sin(2 * np.pi * i) * (0.5 + 0.5 * j) - 4.0 * i
    else:
        lon0 = 1.0e5 * i + 0 * j
        lat0 = 0.6e5 * j + 0 * i
        lon = lon0 + 4.0e3 * np.sin(2 * np.pi * j) * (0.3 + i) + 8.0e3 * j
        lat = lat0 + 3.0e3 * np.sin(2 * np.pi * i) * (0.5 + 0.5 * j) - 5.0e3 * i
    return np.ascontiguousarray(lon), np.ascontiguousarray(lat)


def curv_cgrid_case(name, *, mesh, kernels, seed=0, nx=40, ny=30, nz=5, nt=3, npart=300, field_dtype=np.float32,
                    spatial_dtype="float64", with_w=True, dt=1800.0, runtime=None, vel=0.3, cgrid=True):
    rng = _rng(seed)
    lon, lat = curvilinear_grid(nx, ny, mesh=mesh, seed=seed)
    depth = np.linspace(0, 2000, nz)
    time_s = np.arange(nt) * 86400.0
    shp = (nt, nz, ny, nx)
    fields = {"U": smooth_random_field(rng, shp, vel, field_dtype), "V": smooth_random_field(rng, shp, vel, field_dtype)}
    if cgrid:
        dims = {"U": ("time", "ZC", "YC", "XG"), "V": ("time", "ZC", "YG", "XC")}
    else:
        dims = {"U": TZYX_NODE, "V": TZYX_NODE}
    if with_w:
        fields["W"] = smooth_random_field(rng, shp, 0.01 * vel, field_dtype)
        dims["W"] = ("time", "depth", "YC", "XC") if cgrid else TZYX_NODE
    # release inside the mesh: bilinear blend of interior cells
    ci = rng.uniform(0.2, 0.8, npart) * (nx - 1)
    cj = rng.uniform(0.2, 0.8, npart) * (ny - 1)
    i0, j0 = ci.astype(int), cj.astype(int)
    fi, fj = ci - i0, cj - j0
    def blend(a):
        return (a[j0, i0] * (1 - fi) * (1 - fj) + a[j0, i0 + 1] * fi * (1 - fj) + a[j0 + 1, i0] * (1 - fi) * fj
                + a[j0 + 1, i0 + 1] * fi * fj)
    x, y = blend(lon), blend(lat)
    z = rng.uniform(200, 1800, npart)
    return dict(
        name=name,
        mesh=mesh,
        lon=lon,
        lat=lat,
        depth=depth,
        x_pad="low",
        y_pad="low",
        z_pad="high" if cgrid else "both",
        time_s=time_s,
        fields=fields,
        field_dims=dims,
        cgrid=cgrid,
        kernels=list(kernels),
        spatial_dtype=spatial_dtype,
        x=x,
        y=y,
        z=z,
        t0=None,
        dt=dt,
        runtime=runtime if runtime is not None else 24 * abs(dt),
        seed=seed,
    )


def curv_cgrid_diffusion_case(name, *, mesh, kernels, seed=0, npart=300, spatial_dtype="float64", kh="node4d", dt=1800.0,
                              runtime=None, nx=40, ny=30, field_dtype=np.float32, kh_dtype=np.float32):
    """BASELINE config 5 in small: AdvectionDiffusionM1 / EM (and RK45 lists) on the 3-D curvilinear C-grid, U/V/W staggered,
    Kh_zonal / Kh_meridional with a tanh profile (tests/test_diffusion.py:60-67) living on the NODES of the same grid and
    sampled with XLinear -- either full (time, depth, YG, XG) arrays or 2-D (YG, XG) ones like tools/bench_configs.py."""
    case = curv_cgrid_case(name, mesh=mesh, kernels=kernels, seed=seed, nx=nx, ny=ny, npart=npart, field_dtype=field_dtype,
                           spatial_dtype=spatial_dtype, with_w=True, dt=dt, runtime=runtime, vel=0.3, cgrid=True)
    nt, nz = case["fields"]["U"].shape[:2]
    ii = (np.arange(nx)[None, :] / (nx - 1)) * np.ones((ny, 1))
    jj = (np.arange(ny)[:, None] / (ny - 1)) * np.ones((1, nx))
    scale = 100.0 if mesh == "spherical" else 10.0
    khz = (scale * (1.0 + 0.5 * np.tanh(3 * (2 * ii - 1)))).astype(kh_dtype)
    khm = (scale * (1.0 + 0.3 * np.tanh(2 * (2 * jj - 1)))).astype(kh_dtype)
    if kh == "node4d":
        tz = (1.0 + 0.1 * np.arange(nt))[:, None, None, None] * (1.0 - 0.05 * np.arange(nz))[None, :, None, None]
        case["fields"]["Kh_zonal"] = (tz * khz[None, None]).astype(kh_dtype)
        case["fields"]["Kh_meridional"] = (tz * khm[None, None]).astype(kh_dtype)
        case["field_dims"]["Kh_zonal"] = TZYX_NODE
        case["field_dims"]["Kh_meridional"] = TZYX_NODE
    else:
        case["fields"]["Kh_zonal"] = khz[None, None]
        case["fields"]["Kh_meridional"] = khm[None, None]
        case["field_dims"]["Kh_zonal"] = ("mockT", "mockZ", "YG", "XG")
        case["field_dims"]["Kh_meridional"] = ("mockT", "mockZ", "YG", "XG")
    case["context"] = {"dres": 0.01 if mesh == "spherical" else 100.0}
    case["seed"] = 4321 + seed
    return case


def diffusion_case(name, *, mesh, kernels, seed=0, npart=200, const_kh=None, spatial_dtype="float64", dt=600.0,
                   runtime=6 * 3600.0):
    """Kh fields like tests/test_diffusion.py:49-78 (tanh profile) or constant Kh (:19-46)."""
    rng = _rng(seed)
    nx, ny = 40, 30
    if mesh == "spherical":
        lon = np.linspace(-10, 10, nx)
        lat = np.linspace(-8, 8, ny)
        dres = 0.05
        khscale = 100.0
    else:
        lon = np.linspace(-2.0e4, 2.0e4, nx)
        lat = np.linspace(-1.5e4, 1.5e4, ny)
        dres = 100.0
        khscale = 10.0
    depth = np.array([0.0, 100.0])
    time_s = np.array([0.0, 86400.0])
    shp = (2, 2, ny, nx)
    U = smooth_random_field(rng, shp, 0.05, np.float64)
    V = smooth_random_field(rng, shp, 0.05, np.float64)
    fields = {"U": U, "V": V}
    dims = {"U": TZYX_NODE, "V": TZYX_NODE}
    constants = {}
    if const_kh is not None:
        constants = {"Kh_zonal": const_kh, "Kh_meridional": const_kh}
    else:
        xs = lon / lon[-1]
        khz = khscale * (1 + 0.5 * np.tanh(3 * xs))[None, None, None, :] * np.ones(shp)
        ys = lat / lat[-1]
        khm = khscale * (1 + 0.3 * np.tanh(2 * ys))[None, None, :, None] * np.ones(shp)
        fields["Kh_zonal"] = khz
        fields["Kh_meridional"] = khm
        dims["Kh_zonal"] = TZYX_NODE
        dims["Kh_meridional"] = TZYX_NODE
    x = rng.uniform(lon[0] * 0.3, lon[-1] * 0.3, npart)
    y = rng.uniform(lat[0] * 0.3, lat[-1] * 0.3, npart)
    return dict(
        name=name,
        mesh=mesh,
        lon=lon,
        lat=lat,
        depth=depth,
        x_pad="low",
        y_pad="low",
        z_pad="both",
        time_s=time_s,
        fields=fields,
        field_dims=dims,
        cgrid=False,
        constants=constants,
        const_mesh=mesh,
        context={"dres": dres},
        kernels=list(kernels),
        spatial_dtype=spatial_dtype,
        x=x,
        y=y,
        z=np.full(npart, 10.0),
        t0=None,
        dt=dt,
        runtime=runtime,
        seed=1234 + seed,
    )


def slip_case(name, *, slip, mesh, kernels, seed=0, with_w=False, npart=300, spatial_dtype="float64", field_dtype=np.float64,
              coord_dtype=np.float64):
    """A-grid flow around rectangular "land" blocks (U = V = 0 there) sampled with XFreeslip / XPartialslip
    (tests/test_interpolation.py:119-154 pattern, turned into an advection run)."""
    case = rect_agrid_case(name, mesh=mesh, kernels=kernels, seed=seed, nx=30, ny=20, nz=4, nt=3, npart=npart, with_w=with_w,
                           spatial_dtype=spatial_dtype, field_dtype=field_dtype, coord_dtype=coord_dtype, runtime=20 * 3600.0, wscale=0.002)
    rng = _rng(1000 + seed)
    land = np.zeros((20, 30), bool)
    for _ in range(14):
        j, i = rng.integers(1, 17), rng.integers(1, 26)
        land[j : j + rng.integers(1, 4), i : i + rng.integers(1, 5)] = True
    for f in case["fields"].values():
        f[:, :, land] = 0.0
        f[:, 2:, land | np.roll(land, 1, axis=1)] = 0.0  # wider below: the z1 level is land where z0 is
    case["slip"] = slip
    return case


def slip_curv_case(name, *, slip, mesh, kernels, seed=0, with_w=False, npart=300, spatial_dtype="float64", field_dtype=np.float64,
                   populate=False):
    """XFreeslip / XPartialslip on a CURVILINEAR A-grid with land blocks.  Unpopulated, the first evaluation of the reference
    carries float32 xsi / eta arrays (spatialhash.py:505): f_u = ones_like(xsi) and the slip factors are float32 then."""
    case = curv_cgrid_case(name, mesh=mesh, kernels=kernels, seed=seed, nx=30, ny=20, nz=4, nt=3, npart=npart, field_dtype=field_dtype,
                           spatial_dtype=spatial_dtype, with_w=with_w, dt=1800.0, runtime=12 * 3600.0, vel=0.3, cgrid=False)
    rng = _rng(2000 + seed)
    land = np.zeros((20, 30), bool)
    for _ in range(16):
        j, i = rng.integers(1, 17), rng.integers(1, 26)
        land[j : j + rng.integers(1, 4), i : i + rng.integers(1, 5)] = True
    for f in case["fields"].values():
        f[:, :, land] = 0.0
        f[:, 2:, land | np.roll(land, 1, axis=1)] = 0.0
    case["slip"] = slip
    case["populate"] = populate
    return case


def sample_case(name, *, interp, mesh="flat", seed=0, npts=400, field_dtype=np.float64, zpad="both", uniform_batch=None, curv=False):
    """Field.eval of a scalar field P with one of the scalar interpolators at explicit points (no particles):
    random interior points, exact node hits, points on land blocks (zeros) and points outside the domain."""
    rng = _rng(seed)
    nx, ny, nz, nt = 12, 10, 4, 3
    lon = np.linspace(0, 11, nx) if mesh == "flat" else np.linspace(-20, 24, nx)
    lat = np.linspace(0, 9, ny) if mesh == "flat" else np.linspace(-18, 18, ny)
    depth = np.array([0.0, 10.0, 30.0, 70.0])
    time_s = np.array([0.0, 100.0, 250.0])
    P = (rng.standard_normal((nt, nz, ny, nx)) + 3.0).astype(field_dtype)
    P[:, :, 3:6, 4:7] = 0.0
    P[:, 2:, 6:8, 1:3] = 0.0
    P[1:, :, 1, 9] = 0.0
    t = rng.uniform(0, 250, npts)
    z = rng.uniform(0, 70, npts)
    y = rng.uniform(lat[0], lat[-1], npts)
    x = rng.uniform(lon[0], lon[-1], npts)
    k = npts // 8
    x[:k] = lon[rng.integers(0, nx, k)]          # exact nodes
    y[:k] = lat[rng.integers(0, ny, k)]
    z[k : 2 * k] = depth[rng.integers(0, nz, k)]
    t[2 * k : 3 * k] = time_s[rng.integers(0, nt, k)]
    x[3 * k : 3 * k + 6] = lon[-1] + 0.5          # out of bounds (-> 0)
    y[3 * k + 6 : 3 * k + 12] = lat[0] - 0.5
    if curv:  # the same logical mesh, smoothly warped: Field.eval has no `ei` guess, so xsi / eta come back as float32 arrays
        lon2, lat2 = curvilinear_grid(nx, ny, mesh=mesh, seed=seed)
        ci, cj = rng.uniform(0.02, 0.98, npts) * (nx - 1), rng.uniform(0.02, 0.98, npts) * (ny - 1)
        if mesh == "flat":  # exact nodes (on a spherical mesh the tangent-plane test of a node hit is decided by the last bit of sin/cos)
            ci[:k], cj[:k] = np.round(ci[:k]), np.round(cj[:k])
        i0, j0 = np.minimum(ci.astype(int), nx - 2), np.minimum(cj.astype(int), ny - 2)
        fi, fj = ci - i0, cj - j0
        blend = lambda a: (a[j0, i0] * (1 - fi) * (1 - fj) + a[j0, i0 + 1] * fi * (1 - fj) + a[j0 + 1, i0] * (1 - fi) * fj  # noqa: E731
                           + a[j0 + 1, i0 + 1] * fi * fj)
        x, y = blend(lon2), blend(lat2)
        x[3 * k : 3 * k + 6] = lon2.max() + 5.0  # outside the mesh
        lon, lat = lon2, lat2
    if uniform_batch == "interior":
        # XLinearInvdistLandTracer weights ALL gathered corners alike, so its value depends on the batch-global lenT/lenZ
        # (_xinterpolators.py:575-576); keep the batch uniform: no particle exactly on the first time level / depth
        t[t == 0.0] = time_s[-1]
        z[z == 0.0] = depth[-1]
    elif uniform_batch == "t0":
        t[:] = 0.0  # every particle on the first time level: lenT == 1 for the whole batch
        z[z == 0.0] = depth[-1]
    return dict(
        name=name, kind="sample", mesh=mesh, lon=lon, lat=lat, depth=depth, x_pad="low", y_pad="low", z_pad=zpad, time_s=time_s,
        fields={"P": P}, field_dims={"P": TZYX_NODE}, cgrid=False, scalar_interp={"P": interp}, sample_field="P",
        kernels=[], spatial_dtype="float64", x=x, y=y, z=z, t0=t, dt=1.0, runtime=None, seed=seed,
    )


def all_cases() -> dict:
    """name -> case.  Keep every case small: the fixtures are committed."""
    c = {}

    def add(case):
        c[case["name"]] = case

    # --- rectilinear A-grid (BASELINE config 2 shape, scaled down) ---------------------------------------------
    add(rect_agrid_case("agrid_flat_rk4_f64", mesh="flat", kernels=["AdvectionRK4"], seed=1))
    add(rect_agrid_case("agrid_sph_rk4_f64", mesh="spherical", kernels=["AdvectionRK4"], seed=2))
    add(rect_agrid_case("agrid_sph_rk4_3d_f64", mesh="spherical", kernels=["AdvectionRK4_3D"], seed=3, with_w=True))
    add(rect_agrid_case("agrid_flat_rk4_3d_f64", mesh="flat", kernels=["AdvectionRK4_3D"], seed=4, with_w=True))
    add(rect_agrid_case("agrid_sph_rk4_f32part", mesh="spherical", kernels=["AdvectionRK4"], seed=5, spatial_dtype="float32"))
    add(rect_agrid_case("agrid_flat_rk4_f32part", mesh="flat", kernels=["AdvectionRK4"], seed=6, spatial_dtype="float32"))
    add(rect_agrid_case("agrid_sph_rk4_f32field", mesh="spherical", kernels=["AdvectionRK4"], seed=7, field_dtype=np.float32))
    add(rect_agrid_case("agrid_sph_rk4_f32all", mesh="spherical", kernels=["AdvectionRK4_3D"], seed=8, with_w=True,
                        field_dtype=np.float32, spatial_dtype="float32", coord_dtype=np.float32))
    add(rect_agrid_case("agrid_sph_rk4_backward", mesh="spherical", kernels=["AdvectionRK4"], seed=9, dt=-3600.0,
                        runtime=30 * 3600.0))
    # backward needs particles to start at the end of the interval
    c["agrid_sph_rk4_backward"]["t0"] = np.full(300, 3 * 86400.0)
    add(rect_agrid_case("agrid_sph_rk4_oddstep", mesh="spherical", kernels=["AdvectionRK4"], seed=10, dt=1000.0,
                        runtime=86400.0 + 777.0))
    add(rect_agrid_case("agrid_sph_rk4_stagger", mesh="spherical", kernels=["AdvectionRK4"], seed=11, stagger=True))
    # escaping particles: large velocities, small margin -> out-of-bounds codes (and the silent left/bottom exit)
    add(rect_agrid_case("agrid_flat_rk4_escape", mesh="flat", kernels=["AdvectionRK4"], seed=12, vel=6.0, margin=0.01,
                        dt=1800.0, runtime=20 * 1800.0))
    add(rect_agrid_case("agrid_flat_rk4_3d_escape_delete", mesh="flat", kernels=["AdvectionRK4_3D", "DeleteParticle"],
                        seed=13, vel=6.0, margin=0.01, with_w=True, dt=1800.0, runtime=20 * 1800.0, wscale=0.02))
    add(rect_agrid_case("agrid_sph_rk4_3d_submerge", mesh="spherical", seed=20, with_w=True, wscale=0.05,
                        kernels=["AdvectionRK4_3D", "SubmergeParticle", "DeleteOutOfBounds"]))
    add(rect_agrid_case("agrid_sph_ee", mesh="spherical", kernels=["AdvectionEE"], seed=14))
    add(rect_agrid_case("agrid_sph_rk2", mesh="spherical", kernels=["AdvectionRK2"], seed=15))
    add(rect_agrid_case("agrid_sph_rk2_3d", mesh="spherical", kernels=["AdvectionRK2_3D"], seed=16, with_w=True))
    add(rect_agrid_case("agrid_sph_rk45", mesh="spherical", kernels=["AdvectionRK45"], seed=17, runtime=12 * 3600.0))
    add(rect_agrid_case("agrid_flat_rk45", mesh="flat", kernels=["AdvectionRK45"], seed=18, runtime=12 * 3600.0))
    c["agrid_flat_rk45"]["context"] = {"RK45_tol": 0.5, "RK45_min_dt": 10.0, "RK45_max_dt": 7200.0}
    add(rect_agrid_case("agrid_sph_rk4_outside_time", mesh="spherical", kernels=["AdvectionRK4"], seed=19, nt=2,
                        runtime=30 * 3600.0))

    # --- the toy kernels of the reference's loop tests (tests/common_kernels.py): kernel-order invariance via dx (test_kernel.py:
    #     167-202), misaligned outputdt with a moving particle (test_particlefile.py:331-398), default float32 particles ----------
    add(rect_agrid_case("agrid_flat_move_east_north_f32", mesh="flat", kernels=["MoveEast", "AdvectionRK4", "MoveNorth", "DoNothing"], seed=33,
                        spatial_dtype="float32", vel=0.0, dt=20.0, runtime=100.0))
    c["agrid_flat_move_east_north_f32"]["outputdt"] = 50.0
    add(rect_agrid_case("agrid_sph_do_nothing_backward", mesh="spherical", kernels=["DoNothing"], seed=34, dt=-300.0, runtime=7200.0))
    c["agrid_sph_do_nothing_backward"]["t0"] = np.full(300, 86400.0)
    c["agrid_sph_do_nothing_backward"]["outputdt"] = 3600.0

    # --- the user kernel every tutorial writes: particles.p = fieldset.P[particles] after the advection kernel ----------------
    for nm, kw, pd in (("agrid_sph_rk4_sample_p_f32", dict(mesh="spherical", seed=31), "float32"),
                       ("agrid_flat_rk4_sample_p_f64_escape", dict(mesh="flat", seed=32, vel=6.0, margin=0.01, dt=1800.0, runtime=20 * 1800.0), "float64")):
        cs = rect_agrid_case(nm, kernels=["AdvectionRK4", "SampleP"] + (["DeleteParticle"] if "escape" in nm else []), **kw)
        rng = _rng(kw["seed"] + 1000)
        cs["fields"]["P"] = smooth_random_field(rng, cs["fields"]["U"].shape, 5.0, np.float64)
        cs["field_dims"]["P"] = TZYX_NODE
        cs["sample_into"] = {"SampleP": ["P", "p", pd]}
        add(cs)

    # --- its vector form: particles.u, particles.v = fieldset.UV[particles] (tests/test_particleset_execute.py:195-243): default float32
    #     particles + Variables on a spherical A-grid; (u, _, w) = fieldset.UVW[particles] in float64 on the curvilinear C-grid, escapes
    cs = rect_agrid_case("agrid_sph_rk4_sample_uv_f32", mesh="spherical", kernels=["AdvectionRK4", "SampleUV"], seed=35, spatial_dtype="float32")
    cs["sample_into"] = {"SampleUV": ["UV", ["u", "v"], "float32"]}
    add(cs)
    cs = curv_cgrid_case("cgrid_curv_sph_rk4_3d_sample_uvw", mesh="spherical", kernels=["AdvectionRK4_3D", "SampleUVW", "DeleteParticle"], seed=36, vel=1.0)
    cs["sample_into"] = {"SampleUVW": ["UVW", ["u", None, "w"], "float64"]}
    add(cs)

    # --- scalar interpolators sampled through Field.eval ------------------------------------------------------------
    add(sample_case("sample_xlinear", interp="XLinear", seed=61))
    add(sample_case("sample_xnearest", interp="XNearest", seed=62))
    add(sample_case("sample_cgrid_tracer", interp="CGrid_Tracer", seed=63, zpad="high"))
    add(sample_case("sample_invdist_land", interp="XLinearInvdistLandTracer", seed=64, uniform_batch="interior"))
    add(sample_case("sample_invdist_land_f32", interp="XLinearInvdistLandTracer", seed=65, field_dtype=np.float32, mesh="spherical",
                    uniform_batch="interior"))
    add(sample_case("sample_invdist_land_t0", interp="XLinearInvdistLandTracer", seed=66, uniform_batch="t0"))

    # --- slip boundary conditions (XFreeslip / XPartialslip velocity interpolators) -------------------------------
    add(slip_case("slip_free_flat_rk4", slip="free", mesh="flat", kernels=["AdvectionRK4"], seed=51))
    add(slip_case("slip_partial_sph_rk4", slip="partial", mesh="spherical", kernels=["AdvectionRK4"], seed=52))
    add(slip_case("slip_partial_sph_rk4_3d", slip="partial", mesh="spherical", kernels=["AdvectionRK4_3D", "DeleteParticle"], seed=53,
                  with_w=True))
    add(slip_case("slip_free_sph_rk4_f32", slip="free", mesh="spherical", kernels=["AdvectionRK4"], seed=54,
                  spatial_dtype="float32", field_dtype=np.float32))

    # --- restated analytic datasets (BASELINE config 1) --------------------------------------------------------
    add(peninsula_case("peninsula_A_flat", mesh="flat", grid_type="A"))
    add(peninsula_case("peninsula_A_spherical", mesh="spherical", grid_type="A"))
    add(peninsula_case("peninsula_C_flat", mesh="flat", grid_type="C"))
    add(peninsula_case("peninsula_C_spherical", mesh="spherical", grid_type="C"))
    add(peninsula_case("peninsula_A_flat_f64", mesh="flat", grid_type="A", spatial_dtype="float64"))
    add(stommel_case("stommel_A", grid_type="A"))
    add(stommel_case("stommel_C", grid_type="C"))
    add(moving_eddy_case("moving_eddy_rk4"))
    add(moving_eddy_case("moving_eddy_rk45", kernels=["AdvectionRK45"],
                         context={"RK45_tol": 1e-5, "RK45_min_dt": 1.0, "RK45_max_dt": 3600.0}))
    add(moving_eddy_case("moving_eddy_ee", kernels=["AdvectionEE"]))

    # --- rectilinear C-grid with W -------------------------------------------------------------------------------
    add(rect_cgrid_case("cgrid_rect_flat_rk4_3d", mesh="flat", kernels=["AdvectionRK4_3D"], seed=21))
    add(rect_cgrid_case("cgrid_rect_sph_rk4_3d", mesh="spherical", kernels=["AdvectionRK4_3D"], seed=22))
    add(rect_cgrid_case("cgrid_rect_sph_rk4_f64field", mesh="spherical", kernels=["AdvectionRK4"], seed=23,
                        field_dtype=np.float64, with_w=False))

    # --- curvilinear (spatial hash + tangent-plane point-in-cell) --------------------------------------------------
    add(curv_cgrid_case("cgrid_curv_sph_rk4_3d", mesh="spherical", kernels=["AdvectionRK4_3D", "DeleteParticle"], seed=31))
    add(curv_cgrid_case("cgrid_curv_sph_rk4_3d_err", mesh="spherical", kernels=["AdvectionRK4_3D"], seed=31))
    add(curv_cgrid_case("cgrid_curv_flat_rk4", mesh="flat", kernels=["AdvectionRK4"], seed=32, with_w=False))
    add(curv_cgrid_case("agrid_curv_sph_rk4", mesh="spherical", kernels=["AdvectionRK4", "DeleteParticle"], seed=33,
                        with_w=False, cgrid=False))
    for nm in ("cgrid_curv_sph_rk4_3d", "cgrid_curv_flat_rk4", "agrid_curv_sph_rk4"):
        pc = dict(c[nm])
        pc["name"] = nm + "_populated"
        pc["populate"] = True  # ParticleSet.populate_indices() first: every evaluation has an ei guess
        add(pc)
    add(curv_cgrid_case("cgrid_curv_sph_rk45", mesh="spherical", kernels=["AdvectionRK45"], seed=34, with_w=False,
                        runtime=8 * 3600.0))

    # --- stochastic kernels (counter-based RNG injected into the reference run) ----------------------------------
    add(diffusion_case("diff_m1_flat", mesh="flat", kernels=["AdvectionDiffusionM1"], seed=41))
    add(diffusion_case("diff_m1_sph", mesh="spherical", kernels=["AdvectionDiffusionM1"], seed=42))
    add(diffusion_case("diff_em_sph", mesh="spherical", kernels=["AdvectionDiffusionEM"], seed=43))
    add(diffusion_case("diff_uniform_sph", mesh="spherical", kernels=["AdvectionRK4", "DiffusionUniformKh"], seed=44,
                       const_kh=50.0))
    add(diffusion_case("diff_m1_constkh_flat", mesh="flat", kernels=["AdvectionDiffusionM1"], seed=45, const_kh=5.0))
    add(diffusion_case("diff_m1_sph_f32part", mesh="spherical", kernels=["AdvectionDiffusionM1"], seed=46,
                       spatial_dtype="float32"))

    # --- BASELINE config 5: stochastic + adaptive kernels on the 3-D curvilinear C-grid -------------------------------
    add(curv_cgrid_diffusion_case("cgrid_curv_sph_m1", mesh="spherical", kernels=["AdvectionDiffusionM1", "DeleteParticle"], seed=71))
    add(curv_cgrid_diffusion_case("cgrid_curv_flat_m1", mesh="flat", kernels=["AdvectionDiffusionM1", "DeleteParticle"], seed=72))
    add(curv_cgrid_diffusion_case("cgrid_curv_sph_m1_kh2d", mesh="spherical", kernels=["AdvectionDiffusionM1", "DeleteParticle"], seed=73,
                                  kh="2d"))
    add(curv_cgrid_diffusion_case("cgrid_curv_sph_m1_f32part", mesh="spherical", kernels=["AdvectionDiffusionM1", "DeleteParticle"],
                                  seed=74, spatial_dtype="float32"))
    add(curv_cgrid_diffusion_case("cgrid_curv_sph_em", mesh="spherical", kernels=["AdvectionDiffusionEM", "DeleteParticle"], seed=75,
                                  kh_dtype=np.float64, field_dtype=np.float64))
    pc = dict(curv_cgrid_diffusion_case("cgrid_curv_sph_m1_populated", mesh="spherical", kernels=["AdvectionDiffusionM1", "DeleteParticle"],
                                        seed=76))
    pc["populate"] = True
    add(pc)
    rc = curv_cgrid_case("cgrid_curv_sph_rk45_w", mesh="spherical", kernels=["AdvectionRK45", "DeleteParticle"], seed=77, with_w=True,
                         runtime=8 * 3600.0)
    rc["context"] = {"RK45_tol": 10.0, "RK45_min_dt": 1.0, "RK45_max_dt": 86400.0}  # kernel.py:137-159 defaults, as in config 5
    rc["populate"] = True
    add(rc)
    rc = curv_cgrid_case("cgrid_curv_flat_rk45_f32part", mesh="flat", kernels=["AdvectionRK45", "DeleteParticle"], seed=78, with_w=True,
                         runtime=8 * 3600.0, spatial_dtype="float32")
    rc["context"] = {"RK45_tol": 0.5, "RK45_min_dt": 10.0, "RK45_max_dt": 7200.0}
    add(rc)
    rc = curv_cgrid_case("cgrid_curv_sph_rk45_f32part", mesh="spherical", kernels=["AdvectionRK45", "DeleteParticle"], seed=79, with_w=True,
                         runtime=8 * 3600.0, spatial_dtype="float32")
    rc["context"] = {"RK45_tol": 50.0, "RK45_min_dt": 10.0, "RK45_max_dt": 7200.0}
    rc["next_dt_dtype"] = "float32"
    add(rc)

    # --- float32 ARRAYS meeting in the slip interpolators, XLinearInvdistLandTracer and AdvectionRK45 -----
    add(slip_curv_case("slip_free_curv_sph_f32", slip="free", mesh="spherical", kernels=["AdvectionRK4", "DeleteParticle"], seed=81,
                       spatial_dtype="float32", field_dtype=np.float32))
    add(slip_curv_case("slip_partial_curv_flat_3d", slip="partial", mesh="flat", kernels=["AdvectionRK4_3D", "DeleteParticle"], seed=82,
                       with_w=True))
    add(slip_curv_case("slip_partial_curv_sph_ee", slip="partial", mesh="spherical", kernels=["AdvectionEE", "DeleteParticle"], seed=83,
                       field_dtype=np.float32))
    add(slip_case("slip_partial_sph_f32all_3d", slip="partial", mesh="spherical", kernels=["AdvectionRK4_3D", "DeleteParticle"], seed=84,
                  with_w=True, spatial_dtype="float32", field_dtype=np.float32, coord_dtype=np.float32))
    add(slip_case("slip_free_flat_f32all_ee", slip="free", mesh="flat", kernels=["AdvectionEE"], seed=85, spatial_dtype="float32",
                  field_dtype=np.float32, coord_dtype=np.float32))
    add(sample_case("sample_invdist_land_mixed", interp="XLinearInvdistLandTracer", seed=67))  # nodes at t = 0 and z = 0 among interior points
    add(sample_case("sample_invdist_land_curv_f32", interp="XLinearInvdistLandTracer", seed=68, field_dtype=np.float32, mesh="spherical",
                    curv=True))
    add(sample_case("sample_xlinear_curv_f32", interp="XLinear", seed=69, field_dtype=np.float32, mesh="flat", curv=True))
    add(peninsula_case("peninsula_A_flat_rk45", mesh="flat", grid_type="A", kernels=("AdvectionRK45",), runtime=6 * 3600.0))
    c["peninsula_A_flat_rk45"]["context"] = {"RK45_tol": 1.0, "RK45_min_dt": 10.0, "RK45_max_dt": 3600.0}
    add(stommel_case("stommel_A_rk45", grid_type="A", kernels=("AdvectionRK45",), runtime=5 * 86400.0))
    c["stommel_A_rk45"]["context"] = {"RK45_tol": 100.0, "RK45_min_dt": 60.0, "RK45_max_dt": 86400.0}
    add(peninsula_case("peninsula_C_spherical_rk45", mesh="spherical", grid_type="C", kernels=("AdvectionRK45",), runtime=6 * 3600.0))
    c["peninsula_C_spherical_rk45"]["context"] = {"RK45_tol": 1.0, "RK45_min_dt": 10.0, "RK45_max_dt": 3600.0}
    # --- non-finite release positions (tests/test_spatialhash.py:50-56: NaN/inf -> GRID_SEARCH_ERROR) ---------------
    for nm, base, kern in (("agrid_sph_rk4_nonfinite", "agrid_sph_rk4_f64", ["AdvectionRK4", "DeleteParticle"]),
                           ("cgrid_curv_sph_rk4_nonfinite", "cgrid_curv_sph_rk4_3d", ["AdvectionRK4_3D", "DeleteParticle"])):
        nf = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in c[base].items()}
        nf["name"] = nm
        nf["kernels"] = kern
        nf["x"][3], nf["y"][7], nf["z"][11] = np.nan, np.nan, np.nan
        nf["x"][13], nf["y"][17], nf["z"][19] = np.inf, -np.inf, np.inf
        add(nf)
    nf = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in c["agrid_sph_rk4_f64"].items()}
    nf["name"] = "agrid_sph_rk4_nan_raises"
    nf["x"][5] = np.nan
    add(nf)

    # --- output intervals + the float32 `next_dt` Variable of the reference's tests (tests/utils.py:24-25) -----------
    # outputdt = 2.3 dt: the last step of every interval is clipped to a dt that float32 cannot hold, AdvectionRK45 copies
    # it into next_dt and `dt = next_dt` (kernel.py:118-120) carries the rounded value into the next interval.
    for nm, nd in (("agrid_sph_rk45_outputdt_f64nextdt", "float64"), ("agrid_sph_rk45_outputdt_f32nextdt", "float32")):
        oc = rect_agrid_case(nm, mesh="spherical", kernels=["AdvectionRK45"], seed=62, npart=100, runtime=14 * 3600.0)
        oc["outputdt"] = 2.3 * 3600.0 + 0.7
        oc["next_dt_dtype"] = nd
        oc["context"] = {"RK45_tol": 500.0, "RK45_min_dt": 10.0, "RK45_max_dt": 4 * 3600.0}
        add(oc)
    oc = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in c["agrid_sph_rk45_outputdt_f32nextdt"].items()}
    oc["name"] = "agrid_sph_rk45_outputdt_f32nextdt_maxdt"  # max_dt = dt: the clipped, f32-rounded dt IS the next interval's first step
    oc["context"] = {"RK45_tol": 500.0, "RK45_min_dt": 10.0, "RK45_max_dt": 3600.0}
    add(oc)
    oc = rect_agrid_case("agrid_flat_rk4_3d_outputdt", mesh="flat", kernels=["AdvectionRK4_3D"], seed=63, with_w=True, npart=100,
                         runtime=9 * 3600.0)
    oc["outputdt"] = 2.5 * 3600.0
    add(oc)

    # --- length-1 dimensions (tests/test_advection.py:207-234 test_length1dimensions; index_search.py:45-46) -----
    # A coordinate of one node: index 0, bcoord 0 and no out-of-bounds test on that axis, wherever the particle is.
    for nm, axes in (("agrid_flat_len1_x", "x"), ("agrid_flat_len1_y", "y"), ("agrid_flat_len1_z", "z"), ("agrid_flat_len1_xyz", "xyz")):
        lc = rect_agrid_case(nm, mesh="flat", kernels=["AdvectionRK4_3D"], seed=61, with_w=True, npart=60, runtime=6 * 3600.0)
        for ax, key, axis in (("x", "lon", 3), ("y", "lat", 2), ("z", "depth", 1)):
            if ax in axes:
                lc[key] = lc[key][:1].copy()
                lc["fields"] = {k: np.ascontiguousarray(np.take(v, [0], axis=axis)) for k, v in lc["fields"].items()}
        add(lc)

    # --- errors WITHOUT a recovery kernel: the reference raises after the iteration of its batch loop in which the first particle
    #     errs (kernel.py:236-245), every other particle stopped there too.  Staggered release times (iteration k is every particle's
    #     own k-th step, not a common time), output intervals (the iteration count restarts with every Kernel.execute), the
    #     populated curvilinear C-grid (dedicated kernels) -------------------------------------------------------------------------
    add(rect_agrid_case("agrid_flat_rk4_escape_stagger", mesh="flat", kernels=["AdvectionRK4"], seed=71, vel=5.0, margin=0.01, dt=1800.0,
                        runtime=30 * 1800.0, stagger=True))
    oc = rect_agrid_case("agrid_flat_rk4_3d_escape_outputdt", mesh="flat", kernels=["AdvectionRK4_3D"], seed=72, vel=2.5, margin=0.06, with_w=True,
                         wscale=0.002, dt=1800.0, runtime=40 * 1800.0, npart=200)
    oc["outputdt"] = 3.5 * 1800.0
    add(oc)
    pc = dict(curv_cgrid_case("cgrid_curv_sph_rk4_3d_err_populated_stagger", mesh="spherical", kernels=["AdvectionRK4_3D"], seed=73, vel=1.2))
    pc["populate"] = True
    pc["t0"] = np.round(_rng(74).uniform(0, 6, len(pc["x"]))) * 1800.0
    add(pc)

    # --- 2-D kernels on float32 COORDINATES with float32 particles: `particles.z` is handed to every stage unchanged, so on a float32 depth
    #     axis zeta stays a float32 array in ALL stages (index_search.py:51) while the stage positions y1, x1 are float64 -- found by seed 7163
    #     of tools/fuzz_oracle_vs_reference.py (a float32 ulp in stage 2 that a shear flow grew to 1 m), float64 and float32 field data ------
    add(rect_agrid_case("agrid_flat_rk2_f32coords_f32part", mesh="flat", kernels=["AdvectionRK2", "DeleteParticle"], seed=111, coord_dtype=np.float32,
                        spatial_dtype="float32", stagger=True, vel=2.0, runtime=28 * 3600.0))
    add(rect_agrid_case("agrid_sph_rk4_f32coords_f32part", mesh="spherical", kernels=["AdvectionRK4"], seed=112, coord_dtype=np.float32,
                        spatial_dtype="float32", field_dtype=np.float32))
    rc = rect_agrid_case("agrid_flat_rk45_f32coords_f32part", mesh="flat", kernels=["AdvectionRK45"], seed=113, coord_dtype=np.float32,
                         spatial_dtype="float32", runtime=12 * 3600.0)
    rc["context"] = {"RK45_tol": 0.5, "RK45_min_dt": 10.0, "RK45_max_dt": 7200.0}
    add(rc)

    # --- the call-wide OutsideTimeInterval (index_search.py:85-86 raises for the whole call; field.py:31-44 writes code 70 into EVERY
    #     particle of the view and returns 0): releases staggered by half a step and a run that ends past the last time level, so that in
    #     the iteration in which the first particle leaves the time interval the others are still inside it.  With a recovery kernel
    #     the reference deletes every particle evaluated in that iteration; without one the columns at the raise show it; AdvectionRK45
    #     overwrites the code (_advection.py:146) and goes on with the zeros.  (Found by tools/fuzz_oracle_vs_reference.py, seeds 6102 / 6650.)
    def past_the_end(case, extra_steps=1.0):
        """Stagger the releases by half steps (unless the case already does) and let the run end `extra_steps` steps past the last level."""
        n = len(np.atleast_1d(case["x"]))
        dt = float(case["dt"])
        if case.get("t0") is None:
            case["t0"] = _rng(case["seed"] + 500).integers(0, 6, n) * abs(dt) * 0.5
        tl = float(case["time_s"][-1] - case["time_s"][0])
        if dt < 0:
            case["t0"] = tl - np.asarray(case["t0"])
            case["runtime"] = tl + extra_steps * abs(dt) - float(np.min(tl - case["t0"]))
        else:
            case["runtime"] = tl + extra_steps * abs(dt) - float(np.min(case["t0"]))
        return case

    add(past_the_end(rect_agrid_case("twe_agrid_sph_ee_delete", mesh="spherical", kernels=["AdvectionEE", "DeleteParticle"], seed=91, nt=2,
                                     stagger=True)))
    add(past_the_end(rect_agrid_case("twe_agrid_sph_rk4_raise", mesh="spherical", kernels=["AdvectionRK4"], seed=92, nt=2, stagger=True)))
    add(past_the_end(rect_agrid_case("twe_agrid_flat_rk4_3d_delete_f32part", mesh="flat", kernels=["AdvectionRK4_3D", "DeleteParticle"], seed=93,
                                     nt=2, with_w=True, stagger=True, spatial_dtype="float32")))
    add(past_the_end(rect_agrid_case("twe_agrid_sph_rk2_3d_oob_raise", mesh="spherical", kernels=["AdvectionRK2_3D", "DeleteOutOfBounds"], seed=94,
                                     nt=2, with_w=True, stagger=True)))
    add(past_the_end(rect_agrid_case("twe_agrid_sph_rk4_backward_delete", mesh="spherical", kernels=["AdvectionRK4", "DeleteParticle"], seed=95,
                                     nt=2, stagger=True, dt=-3600.0)))
    rc = past_the_end(rect_agrid_case("twe_agrid_sph_rk45", mesh="spherical", kernels=["AdvectionRK45"], seed=96, nt=2, stagger=True), 2.0)
    rc["context"] = {"RK45_tol": 500.0, "RK45_min_dt": 10.0, "RK45_max_dt": 7200.0}
    add(rc)
    oc = past_the_end(rect_agrid_case("twe_agrid_sph_rk4_outputdt_delete", mesh="spherical", kernels=["AdvectionRK4", "DeleteParticle"], seed=97,
                                      nt=2, stagger=True, npart=100))
    oc["outputdt"] = 7.5 * 3600.0
    add(oc)
    cs = past_the_end(rect_agrid_case("twe_agrid_sph_rk4_sample_p_delete", mesh="spherical", kernels=["AdvectionRK4", "SampleP", "DeleteParticle"],
                                      seed=98, nt=2, stagger=True))
    cs["fields"]["P"] = smooth_random_field(_rng(1098), cs["fields"]["U"].shape, 5.0, np.float64)
    cs["field_dims"]["P"] = TZYX_NODE
    cs["sample_into"] = {"SampleP": ["P", "p", "float64"]}
    add(cs)
    # the dedicated curvilinear C-grid kernels (populated: every evaluation has a guess): RK4_3D, RK45, M1
    for nm, kern, seed, kw in (("twe_cgrid_curv_sph_rk4_3d_delete", ["AdvectionRK4_3D", "DeleteParticle"], 101, {}),
                               ("twe_cgrid_curv_sph_rk4_3d_raise", ["AdvectionRK4_3D"], 102, {"vel": 0.03}),  # (slow: nobody leaves the mesh first)
                               ("twe_cgrid_curv_sph_rk45_delete", ["AdvectionRK45", "DeleteParticle"], 103, {})):
        pc = dict(curv_cgrid_case(nm, mesh="spherical", kernels=kern, seed=seed, nt=2, **kw))
        pc["populate"] = True
        pc["t0"] = np.round(_rng(seed + 500).uniform(0, 5, len(pc["x"]))) * 900.0
        if "AdvectionRK45" in kern:
            pc["context"] = {"RK45_tol": 10.0, "RK45_min_dt": 1.0, "RK45_max_dt": 86400.0}
        add(past_the_end(pc, 2.0 if "AdvectionRK45" in kern else 1.0))
    pc = dict(curv_cgrid_diffusion_case("twe_cgrid_curv_sph_m1_delete", mesh="spherical", kernels=["AdvectionDiffusionM1", "DeleteParticle"], seed=104))
    pc["populate"] = True
    pc["t0"] = np.round(_rng(604).uniform(0, 5, len(pc["x"]))) * 900.0
    add(past_the_end(pc))

    # --- SEVERAL execute() calls on one ParticleSet (the usual script loop; parcels_amd keeps the columns on the device between the calls):
    #     every call sets dt anew (particleset.py:381), takes its start time from the particles (:523-585) and builds a Kernel (RK45_tol is
    #     divided by deg2m again on a spherical mesh, kernel.py:144-145); next_dt, the positions and `ei` carry over, deleted particles stay away
    mc = rect_agrid_case("multi_agrid_sph_rk4_three_calls", mesh="spherical", kernels=["AdvectionRK4", "DeleteParticle"], seed=121, vel=6.0, margin=0.02,
                         runtime=7 * 3600.0)
    mc["more_calls"] = [{"dt": 1800.0, "runtime": 5 * 3600.0}, {"dt": 3600.0, "runtime": 6 * 3600.0}]
    add(mc)
    mc = rect_agrid_case("multi_agrid_flat_rk45_two_calls", mesh="flat", kernels=["AdvectionRK45"], seed=122, runtime=6 * 3600.0)
    mc["context"] = {"RK45_tol": 5.0, "RK45_min_dt": 10.0, "RK45_max_dt": 7200.0}
    mc["more_calls"] = [{"dt": 900.0, "runtime": 6 * 3600.0}]
    add(mc)
    mc = dict(curv_cgrid_case("multi_cgrid_curv_sph_rk45_two_calls", mesh="spherical", kernels=["AdvectionRK45", "DeleteParticle"], seed=123, with_w=False,
                              runtime=8 * 3600.0, vel=1.5))
    mc["populate"] = True
    mc["context"] = {"RK45_tol": 30.0, "RK45_min_dt": 60.0, "RK45_max_dt": 4 * 3600.0}
    mc["more_calls"] = [{"dt": 1800.0, "runtime": 8 * 3600.0}]
    add(mc)
    mc = dict(curv_cgrid_case("multi_cgrid_curv_sph_rk4_3d_backward_then_forward", mesh="spherical", kernels=["AdvectionRK4_3D", "DeleteParticle"], seed=124,
                              dt=-1800.0, runtime=6 * 3600.0, vel=1.0))
    mc["populate"] = True
    mc["t0"] = 12 * 3600.0
    mc["more_calls"] = [{"dt": 1800.0, "runtime": 9 * 3600.0}]
    add(mc)

    # --- the `_delete` cases above end with NO particle left (everyone evaluated in the fatal iteration is deleted): they pin the
    #     discrete outcome but not a single position.  Siblings with an output interval: the positions, ids and times of the doomed
    #     particles at every output time before the sample that fails call-wide (and the interval structure: one Kernel.execute per
    #     interval, the keys of the failing samples counted per call) are compared too.
    for nm in ("twe_agrid_sph_ee_delete", "twe_agrid_flat_rk4_3d_delete_f32part", "twe_agrid_sph_rk4_backward_delete",
               "twe_agrid_sph_rk4_sample_p_delete", "twe_cgrid_curv_sph_rk4_3d_delete", "twe_cgrid_curv_sph_m1_delete"):
        import copy

        sib = copy.deepcopy(c[nm])
        sib["name"] = nm + "_outputdt"
        sib["outputdt"] = 5.0 * 3600.0
        add(sib)

    return c
