"""CPU: host-side logic of parcels_amd (no GPU, no compute calls)."""

import ctypes as C
import os
import re

import numpy as np
import pytest

import parcels_amd as pa
from case_utils import ROOT_DIR, build_fieldset, build_pset, golden_names, is_curvilinear, load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from parcels_amd import _hip

    lib = _hip.load()
    header = open(os.path.join(ROOT, "include", "parcels_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(pk_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations found in include/parcels_hip.h"
    assert sorted(_hip.ABI_SYMBOLS) == declared
    for sym in declared:
        assert hasattr(lib, sym), f"libparcels_hip.so does not export {sym}"
    assert lib.pk_abi_version() == _hip.PK_ABI_VERSION == 9


def test_ctypes_structs_match_header_layout(tmp_path):
    """sizeof() of every struct of include/parcels_hip.h as gcc lays it out == the ctypes mirror in parcels_amd/_hip.py."""
    import subprocess

    from parcels_amd import _hip

    names = {"pk_grid_desc": _hip.GridDesc, "pk_field_desc": _hip.FieldDesc, "pk_particles_desc": _hip.ParticlesDesc,
             "pk_exec_params": _hip.ExecParams, "pk_exec_stats": _hip.ExecStats, "pk_device_info": _hip.DeviceInfo, "pk_hash_info": _hip.HashInfo}
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "parcels_hip.h"\nint main(void){' +
                   "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT_DIR, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for n, cls in names.items():
        assert C.sizeof(cls) == int(out[n]), n
    # the members the kernels index by constant
    assert _hip.PK_MAX_EXTRA == 8 and _hip.PK_MAX_KERNELS == 8


def test_no_gpu_fails_loudly():
    """The product path has no CPU fallback: without a device the engine must raise, not compute."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    case, _, _ = load_golden("agrid_flat_rk4_f64")
    fs = build_fieldset(case)
    pset = build_pset(case, fs)
    with pytest.raises(pa._hip.HipLibraryError):
        pset.execute(pa.AdvectionRK4, dt=3600.0, runtime=3600.0)


def test_python_kernels_are_accepted_but_still_need_the_device():
    """A Python kernel puts the loop of Kernel.execute on the host (parcels_amd/hostkernels.py); the built-in kernels of the list and
    all field sampling stay on the GPU, so without a device the run fails as loudly as any other."""
    case, _, _ = load_golden("agrid_flat_rk4_f64")
    fs = build_fieldset(case)
    pset = build_pset(case, fs)

    def MyKernel(particles, fieldset):
        particles.dx += 1

    k = pa.Kernel([pa.AdvectionRK4, MyKernel], pset)
    assert k.host_functions == ["MyKernel"] and k.kernel_ids[0] is not None and k.kernel_ids[1] is None
    with pytest.raises(pa._hip.HipLibraryError):
        pset.execute([pa.AdvectionRK4, MyKernel], dt=3600.0, runtime=3600.0)

    def BadSignature(p):
        pass

    with pytest.raises(ValueError):
        pset.execute(BadSignature, dt=3600.0, runtime=3600.0)
    with pytest.raises(RuntimeError):
        pa.AdvectionRK4(None, None)  # device kernels cannot run on the host


@pytest.mark.parametrize("name", ["agrid_sph_rk4_3d_f64", "cgrid_rect_sph_rk4_3d", "peninsula_C_flat", "cgrid_curv_flat_rk4", "diff_uniform_sph"])
def test_grid_metadata_matches_oracle_marshalling(name):
    """axes / ravel dims / C-grid offsets derived by XGrid agree with the oracle's independent derivation."""
    from oracle import c_oracle as co

    case, _, _ = load_golden(name)
    fs = build_fieldset(case)
    g = fs.gridset[0]
    meta = co.grid_meta(case)
    assert [int(a in g.axes) for a in "XYZ"] == [meta["has_x"], meta["has_y"], meta["has_z"]]
    for ax in g.axes:
        assert g.get_axis_dim(ax) == meta[ax.lower() + "dim"]
    off = g.offsets()
    assert (off["X"], off["Y"]) == (meta["off_x"], meta["off_y"])
    if "Z" in g.axes:
        assert off["Z"] == meta["off_z"]
    assert g.is_curvilinear == is_curvilinear(case)
    uv = fs.UV
    assert isinstance(uv.interp_method, pa.CGrid_Velocity if case.get("cgrid") else pa.XLinear_Velocity)
    assert len(fs.gridset) == (2 if case.get("constants") else 1)


def test_tzyx_transposition_and_nan_fill():
    from parcels_amd.field import transpose_to_tzyx

    md = pa.SGrid2DMetadata(node_dimensions=("XG", "YG"),
                            face_dimensions=(pa.FaceNodePadding("XC", "XG", pa.Padding.LOW), pa.FaceNodePadding("YC", "YG", pa.Padding.LOW)),
                            vertical_dimensions=(pa.FaceNodePadding("ZC", "depth", pa.Padding.BOTH),))
    a = np.arange(2 * 3 * 4.0).reshape(4, 3, 2)  # (XG, YG, time)
    da = transpose_to_tzyx(pa.DataArray(("XG", "YG", "time"), a), md)
    assert da.dims == ("time", "mockZ", "YG", "XG") and da.shape == (2, 1, 3, 4)
    assert da.data[1, 0, 2, 3] == a[3, 2, 1]
    u = np.ones((3, 4))
    u[1, 1] = np.nan
    ds = pa.Dataset({"U": (("YG", "XG"), u), "V": (("YG", "XG"), np.ones((3, 4)))},
                    {"lon": (("XG",), np.arange(4.0)), "lat": (("YG",), np.arange(3.0))}, sgrid=md)
    fs = pa.FieldSet.from_sgrid_conventions(ds, mesh="flat")
    assert fs.U.data.data[0, 0, 1, 1] == 0.0  # model.py:135-143 fillna(0)
    assert fs.time_interval is None
    with pytest.raises(AttributeError):
        fs.nonexistent
    fs.add_context("dres", 0.1)
    assert fs.dres == 0.1
    with pytest.raises(AttributeError):
        fs.dres = 3


def test_rk45_context_defaults_follow_reference():
    """kernel.py:134-159: defaults are installed on Kernel construction, tol divided by deg2m on a spherical mesh."""
    case, _, _ = load_golden("agrid_sph_rk45")
    fs = build_fieldset(case)
    pset = build_pset(case, fs)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pa.Kernel([pa.AdvectionRK45], pset)
    assert fs.RK45_min_dt == 1 and fs.RK45_max_dt == 86400
    assert fs.RK45_tol == pytest.approx(10 / (6366707.019493707 * np.pi / 180))
    pclass = pa.get_default_particle(np.float64)
    p2 = pa.ParticleSet(fs, pclass=pclass, x=[1.0], y=[1.0], z=[1.0], t=[0.0])
    with pytest.raises(ValueError):
        pa.Kernel([pa.AdvectionRK45], p2)  # needs next_dt


def test_default_particle_schema():
    p = pa.Particle
    d = {v.name: v for v in p.variables}
    assert d["x"].dtype == np.float32 and d["t"].dtype == np.float64 and d["state"].dtype == np.int32
    assert d["dt"].initial == 1.0 and d["state"].initial == pa.StatusCode.Evaluate
    assert [v.name for v in p.variables if not v.to_write] == ["dz", "dy", "dx", "dt", "state"]


def test_spatial_hash_build_pinned_to_reference():
    from case_utils import attach_hash_table

    for name in golden_names():
        case, _, _ = load_golden(name)
        if is_curvilinear(case) and "hash_checksum" in case:
            attach_hash_table(case)  # asserts the SHA-256 of keys/starts/counts/faces and bitwidth/bbox
            break
    else:
        pytest.fail("no curvilinear fixture")


def test_nemo_to_sgrid_names_dims_offsets_and_w_sign():
    """parcels_amd.convert.nemo_to_sgrid against what the reference's converter produces (convert.py:308-408; expectations
    of tests/test_convert.py:26-104): SGRID topology, U on (y_center, x), V on (y, x_center), offsets X=1, Y=1, Z=0, W negated,
    NEMO names mapped."""
    from parcels_amd import convert

    nt, nz, ny, nx = 2, 3, 6, 7
    rng = np.random.default_rng(0)
    glamf = np.linspace(0, 6, nx)[None, :] + np.zeros((ny, 1))
    gphif = np.linspace(40, 45, ny)[:, None] + np.zeros((1, nx))
    uo, vo, wo = (rng.standard_normal((nt, nz, ny, nx)).astype(np.float32) for _ in range(3))
    coords = pa.Dataset({}, {"glamf": (("t", "y", "x"), glamf[None]), "gphif": (("t", "y", "x"), gphif[None]),
                             "depthw": (("depthw",), np.array([0.0, 10.0, 30.0])), "time_counter": (("time_counter",), np.array([0.0, 86400.0]))})
    ds = convert.nemo_to_sgrid(
        fields={"uo": (("time_counter", "depthu", "y", "x"), uo), "vo": (("time_counter", "depthv", "y", "x"), vo),
                "wo": (("time_counter", "depthw", "y", "x"), wo)}, coords=coords)
    md = ds.sgrid
    assert md.node_dimensions == ("x", "y") and md.node_coordinates == ("lon", "lat")
    assert [(f.face, f.node, f.padding) for f in md.face_dimensions] == [("x_center", "x", pa.Padding.LOW), ("y_center", "y", pa.Padding.LOW)]
    assert [(f.face, f.node, f.padding) for f in md.vertical_dimensions] == [("depth_center", "depth", pa.Padding.HIGH)]
    assert ds["U"].dims == ("time", "depth_center", "y_center", "x")
    assert ds["V"].dims == ("time", "depth_center", "y", "x_center")
    assert ds["W"].dims == ("time", "depth", "y", "x")
    assert np.array_equal(ds["W"].data, -wo) and np.array_equal(ds["U"].data, uo)
    assert ds["lon"].dims == ("y", "x") and ds["lon"].attrs["units"] == "degrees"
    fs = pa.FieldSet.from_sgrid_conventions(ds)  # mesh from the units attribute: spherical
    assert fs.gridset[0]._mesh.is_spherical()
    assert fs.gridset[0].offsets() == {"X": 1, "Y": 1, "Z": 0}
    assert isinstance(fs.UVW.interp_method, pa.CGrid_Velocity)

    # surface-only data: a single depth level is added (convert.py:150-154); a missing coordinate is an error
    ds2 = convert.nemo_to_sgrid(fields={"U": (("time_counter", "y", "x"), uo[:, 0]), "V": (("time_counter", "y", "x"), vo[:, 0])},
                                coords=pa.Dataset({}, {"glamf": (("y", "x"), glamf), "gphif": (("y", "x"), gphif),
                                                       "time_counter": (("time_counter",), np.array([0.0, 86400.0]))}))
    assert ds2["U"].dims == ("time", "depth", "y_center", "x") and ds2["depth"].data.tolist() == [0.0]
    pa.FieldSet.from_sgrid_conventions(ds2, mesh="spherical")
    with pytest.raises(ValueError, match="glamf"):
        convert.nemo_to_sgrid(fields={}, coords=pa.Dataset({}, {"gphif": (("y", "x"), gphif)}))


@pytest.mark.parametrize("mesh,nx,ny", [("spherical", 60, 45), ("flat", 60, 45), ("spherical", 24, 18)])
def test_spatial_hash_build_equals_reference_on_masked_meshes(mesh, nx, ny):
    """Warped meshes with NaN (masked) nodes, fine and coarse (bit width bisected): parcels_amd.spatialhash builds the table
    the reference's SpatialHash builds (spatialhash.py:45-387), array for array.  The GPU build is compared with this one in
    tests/test_gpu_parity.py.  Needs the reference tree (build container only)."""
    from oracle import ref_shim as rs

    if not rs.reference_available():
        pytest.skip("the reference tree is not available on this machine")
    from parcels_amd import spatialhash as sh

    i = np.arange(nx, dtype=np.float64)[None, :] / (nx - 1)
    j = np.arange(ny, dtype=np.float64)[:, None] / (ny - 1)
    lon = -170.0 + 340.0 * i + 2.0 * np.sin(2 * np.pi * j) * (0.3 + i) + 3.0 * j
    lat = -75.0 + 150.0 * j + 1.5 * np.sin(2 * np.pi * i) * (0.5 + 0.5 * j) - 2.0 * i
    js, is_ = slice(ny // 5, ny // 5 + max(ny // 50, 1)), slice(nx // 5, nx // 5 + max(nx // 40, 1))
    lon[js, is_] = np.nan
    lat[js, is_] = np.nan
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = rs.make_ref_grid(lon=lon, lat=lat, depth=None, mesh=mesh).get_spatial_hash()
        mine = sh.SpatialHash(lon, lat, mesh == "spherical").table()
    assert int(ref._bitwidth) == mine["bitwidth"]
    for k in ("keys", "starts", "counts", "faces"):
        assert np.array_equal(np.asarray(ref._hash_table[k]).astype(mine[k].dtype), mine[k]), k


def test_start_and_end_times_follow_the_reference():
    """_get_simulation_start_and_end_times (particleset.py:523-585): min()/max() of the release times propagate NaN (one unset
    time => the run starts at the fieldset start), negative runtime raises, release times outside the time interval and output
    intervals that do not divide the release offsets warn."""
    import warnings

    from parcels_amd.particleset import ParticleSetWarning, _warn_outputdt_release_desync

    case, _, _ = load_golden("agrid_flat_rk4_f64")
    fs = build_fieldset(case)
    n = len(case["x"])
    t = np.linspace(1000.0, 5000.0, n)
    pset = pa.ParticleSet(fs, pclass=pa.get_default_particle(np.float64), x=case["x"], y=case["y"], z=case["z"], t=t)
    assert pset._start_and_end_times(3600.0, None, 1) == (1000.0, 4600.0)
    assert pset._start_and_end_times(3600.0, None, -1) == (5000.0, 1400.0)
    pset._data["t"][3] = np.nan  # the reference: first_release_time = release_times.min() -> NaN -> fieldset start
    tlen = fs.time_interval.time_length_as_flt
    assert pset._start_and_end_times(3600.0, None, 1) == (0.0, 3600.0)
    assert pset._start_and_end_times(3600.0, None, -1) == (tlen, tlen - 3600.0)
    with pytest.raises(ValueError):
        pset.execute(pa.AdvectionRK4, dt=3600.0, runtime=-1.0)
    with pytest.raises(ValueError):
        pset.execute(pa.AdvectionRK4, dt=3600.0, runtime=10.0, endtime=20.0)
    with pytest.warns(ParticleSetWarning):
        pa.ParticleSet(fs, x=case["x"], y=case["y"], z=case["z"], t=np.full(n, -5.0))
    with pytest.warns(ParticleSetWarning):
        _warn_outputdt_release_desync(600.0, 0.0, np.array([0.0, 900.0]))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        _warn_outputdt_release_desync(600.0, 0.0, np.array([0.0, 1200.0, np.nan]))
        _warn_outputdt_release_desync(None, 0.0, np.array([0.0, 900.0]))


def test_changing_an_interpolator_drops_the_device_copy():
    """The scalar interpolator code and the C-grid packing are frozen into the device descriptors: a later assignment must not
    be silently ignored (the FieldSet forgets its engine and rebuilds it on the next use)."""
    case, _, _ = load_golden("agrid_flat_rk4_f64")
    fs = build_fieldset(case)
    fs.__dict__["_engine"] = object()
    fs.U.interp_method = pa.XNearest()
    assert fs._engine is None
    fs.__dict__["_engine"] = object()
    fs.UV.interp_method = pa.XFreeslip()
    assert fs._engine is None


def test_sample_field_token_validation():
    """SampleField(field, into=variable): the device form of `particles.p = fieldset.P[particles]` (tests/test_particleset_execute.py:
    182-205 pattern) -- checked at Kernel construction like the reference checks its kernels (kernel.py:67-70,122-159)."""
    from parcels_amd.kernel import Kernel

    case, _, _ = load_golden("agrid_sph_rk4_sample_p_f32")
    fs = build_fieldset(case)
    pset = build_pset(case, fs)
    k = Kernel([pa.AdvectionRK4, pa.SampleField("P", into="p")], pset)
    assert k.kernel_ids == [4, 10] and k.samples == {1: ("P", 0)} and k.device_variables == ["p"] and k.funcname == "AdvectionRK4SampleP"
    with pytest.raises(ValueError):
        Kernel([pa.SampleField("nope", into="p")], pset)
    with pytest.raises(ValueError):
        Kernel([pa.SampleField("P", into="q")], pset)  # no such Variable
    with pytest.raises(ValueError):
        Kernel([pa.SampleField("P", into="x")], pset)  # a built-in column is not a user Variable
    with pytest.raises(ValueError):
        Kernel([pa.SampleField("UV", into="p")], pset)  # vector fields are sampled by the advection kernels
    with pytest.warns(RuntimeWarning, match="Sampling of velocities should normally be done using fieldset.UV"):
        Kernel([pa.SampleField("U", into="p")], pset)  # field.py:187-190
    ip = pa.ParticleSet(fs, pclass=pa.get_default_particle(np.float64).add_variable(pa.Variable("p", dtype=np.int32)), x=[1.0], y=[1.0], z=[1.0], t=[0.0])
    with pytest.raises(TypeError):
        Kernel([pa.SampleField("P", into="p")], ip)
    with pytest.raises(RuntimeError):
        pa.SampleField("P", into="p")(None, None)  # device kernels cannot run on the host
    # vector form: particles.u, particles.v = fieldset.UV[particles] (tests/test_particleset_execute.py:195-243)
    case, _, _ = load_golden("cgrid_curv_sph_rk4_3d_sample_uvw")
    fs = build_fieldset(case)
    pset = build_pset(case, fs)
    k = Kernel([pa.AdvectionRK4_3D, pa.SampleField("UVW", into=("u", None, "w"))], pset)
    assert k.samples == {1: ("UVW", 0 | 0xFF << 8 | 1 << 16)} and k.device_variables == ["u", "w"] and k.funcname == "AdvectionRK4_3DSampleUVW"
    assert Kernel([pa.SampleField("UV", into=("w", "u"))], pset).samples == {0: ("UV", 0 | 1 << 8)}
    with pytest.raises(ValueError):
        Kernel([pa.SampleField("UV", into=("u", None, "w"))], pset)  # UV has two components
    with pytest.raises(ValueError):
        Kernel([pa.SampleField("W", into=("u", "w"))], pset)  # a tuple needs a vector field
    with pytest.raises(TypeError):
        pa.SampleField("UV", into=(None, None))


def test_host_staging_fills_match_the_plain_loop():
    """pk_host_stage.cpp: the AVX2 / pooled {U,V[,W]} interleave of the level stream equals the scalar loop for every variant
    (runs on the host: no device needed)."""
    from parcels_amd import _hip

    lib = _hip.load()
    assert lib.pk_host_stage_selftest() == 0


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under parcels_amd/ or include/ may import, link or name it (a comment in pk_device.h says
    exactly that), bench.py only in its cpu_baseline leg, __graft_entry__ only to build it and in smoke()."""
    import ast

    prod = os.path.join(ROOT, "parcels_amd")
    for dirpath, _, files in os.walk(prod):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py"):
                tree = ast.parse(open(path).read())
                for node in ast.walk(tree):
                    if isinstance(node, ast.Import):
                        assert not any(a.name.split(".")[0] == "oracle" for a in node.names), path
                    if isinstance(node, ast.ImportFrom):
                        assert (node.module or "").split(".")[0] != "oracle", path
            elif f.endswith((".h", ".hip", ".cpp")) or f == "Makefile":
                for ln in open(path, errors="ignore"):
                    if "oracle" in ln:
                        assert "shares no code with oracle/" in ln, (path, ln)
    for ln in open(os.path.join(ROOT, "include", "parcels_hip.h")):
        assert "oracle" not in ln
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        imports = [n for n in ast.walk(fn) if isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "oracle"]
        if imports:
            assert fn.name == "cpu_baseline", fn.name


def test_bench_gpus_flag_decides_the_world_size():
    """`python bench.py --gpus N` must run N ranks: it launches them itself when no launcher set WORLD_SIZE, and refuses a mismatch."""
    import subprocess
    import sys

    import bench

    assert bench.launch_command_world(8, {}) == (8, True)
    assert bench.launch_command_world(8, {"WORLD_SIZE": "8"}) == (8, False)
    assert bench.launch_command_world(1, {}) == (1, False)
    # a launcher that started fewer ranks than --gpus asks for is an error, not a silent n_gpus = 1 line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_integration_notes_name_every_entry_point():
    """INTEGRATION.md maps the C ABI to the reference interfaces it replaces: no exported function may be missing from it."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "parcels_hip.h")).read()
    notes = open(os.path.join(root, "INTEGRATION.md")).read()
    names = set(re.findall(r"^(?:int32_t|void|double|const char\*)\s+\*?(pk_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert len(names) > 30
    missing = sorted(n for n in names if n not in notes)
    assert not missing, missing


def test_fieldset_describe_and_windowed_arrays():
    """fieldset.py:142-173, 315-330: to_windowed_arrays is a request for the device ring (no device needed to make it, idempotent, chains),
    describe writes one row per field / vector field / context value plus mesh and time interval."""
    import io

    from case_utils import build_fieldset
    from oracle import cases

    case = cases.rect_agrid_case("desc", mesh="spherical", kernels=["AdvectionRK4"], seed=1, npart=4, nx=6, ny=5, nz=2, nt=3)
    fs = build_fieldset(case)
    fs.add_constant_field("Kh_zonal", 10.0, mesh="spherical")
    fs.add_context("max_age", 3.5)
    assert fs.to_windowed_arrays() is fs and fs.__dict__["_window_slots"] == 3
    assert fs.to_windowed_arrays(max_levels=2).__dict__["_window_slots"] == 3 and fs.to_windowed_arrays(max_levels=5).__dict__["_window_slots"] == 5
    with pytest.raises(ValueError):
        fs.to_windowed_arrays(max_levels=0)
    buf = io.StringIO()
    fs.describe(buf)
    text = buf.getvalue()
    lines = text.splitlines()
    assert lines[0].split("|")[1].strip() == "Name" and "Parcels backend" in lines[0]
    names = [ln.split("|")[1].strip() for ln in lines[2:] if ln.startswith("|")]
    assert set(names) == {"U", "V", "UV", "Kh_zonal", "max_age"}
    row = {ln.split("|")[1].strip(): [c.strip() for c in ln.split("|")[2:-1]] for ln in lines[2:] if ln.startswith("|")}
    assert row["U"][0] == "Field" and row["UV"][0] == "VectorField" and row["max_age"] == ["Context", "-", "3.5", "-"]
    assert row["U"][3] == "NumPy" and row["Kh_zonal"][1] != row["U"][1]  # the constant field lives on its own 1 x 1 grid
    assert "mesh: " in text and "time interval: " in text


def test_counters_are_stale_by_machine_code_not_by_source_text(tmp_path, monkeypatch):
    """bench.py marks the committed PMC counters of a kernel stale when the library's kernel has OTHER MACHINE CODE than the profiled one
    (tools/kernel_code_hash.py, written next to the library by make); without a code hash on either side, by the hash of the sources."""
    import json

    import bench

    cur = os.path.join(bench.ROOT, "parcels_amd", "kernel_code_hashes.json")
    if not os.path.exists(cur):
        pytest.skip("no kernel_code_hashes.json (library not built with the ROCm LLVM tools)")
    hashes = json.load(open(cur))
    assert set(hashes) == {"AdvectionRK4", "AdvectionRK4_3D", "AdvectionRK45", "AdvectionDiffusionM1"}
    for key, v in hashes.items():
        assert len(v["code_hash"]) == 16 and v["code_bytes"] > 1000 and "advect" in v["kernel"]
        assert bench.counters_stale(key, {"code_hash": v["code_hash"], "source_hash": "something else"}) is False
        assert bench.counters_stale(key, {"code_hash": "0" * 16, "source_hash": bench.kernel_source_hash()}) is True
        assert bench.counters_stale(key, {"source_hash": bench.kernel_source_hash()}) is False  # (old summaries: sources decide)
        assert bench.counters_stale(key, {"source_hash": "something else"}) is True
