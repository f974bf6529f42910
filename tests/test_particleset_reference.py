"""The reference's tests/test_particleset.py restated against parcels_amd (construction, ids, user Variables, add / merge / remove,
iteration, default depth, populate_indices).  Tests that run the time loop use the native DoNothing token and need the GPU."""

from operator import attrgetter

import numpy as np
import pytest

import parcels_amd as pa
from test_particlefile_reference import make_fieldset


@pytest.fixture
def fieldset():
    return make_fieldset()


def test_pset_create_lon_lat(fieldset):  # test_particleset.py:21-27
    npart = 100
    lon = np.linspace(0, 1, npart, dtype=np.float32)
    lat = np.linspace(1, 0, npart, dtype=np.float32)
    pset = pa.ParticleSet(fieldset, x=lon, y=lat, pclass=pa.Particle)
    assert np.allclose([p.x for p in pset], lon, rtol=1e-12)
    assert np.allclose([p.y for p in pset], lat, rtol=1e-12)


def test_create_empty_pset(fieldset):  # :30-35 (an empty set returns before anything is validated or launched)
    pset = pa.ParticleSet(fieldset, pclass=pa.Particle)
    assert pset.size == 0
    pset.execute(pa.DoNothing, endtime=1.0, dt=1.0)
    assert pset.size == 0


@pytest.mark.parametrize("offset", [0, 1, 200])
def test_pset_with_pids(fieldset, offset, npart=100):  # :38-44
    ids = np.arange(offset, npart + offset)
    pset = pa.ParticleSet(fieldset, x=np.linspace(0, 1, npart), y=np.linspace(1, 0, npart), particle_ids=ids)
    assert np.allclose([p.particle_id for p in pset], ids, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("aslist", [True, False])
def test_pset_customvars_on_pset(gpu, fieldset, aslist):  # :47-59
    if aslist:
        MyParticle = pa.Particle.add_variable([pa.Variable("sample_var"), pa.Variable("sample_var2")])
        pset = pa.ParticleSet(fieldset, x=0, y=0, pclass=MyParticle, sample_var=5.0, sample_var2=10.0)
    else:
        MyParticle = pa.Particle.add_variable(pa.Variable("sample_var"))
        pset = pa.ParticleSet(fieldset, x=0, y=0, pclass=MyParticle, sample_var=5.0)
    pset.execute(pa.DoNothing, dt=np.timedelta64(1, "s"), runtime=np.timedelta64(21, "s"))
    assert np.allclose([p.sample_var for p in pset], 5.0)
    if aslist:
        assert np.allclose([p.sample_var2 for p in pset], 10.0)


@pytest.mark.gpu
def test_pset_custominit_on_pset_attrgetter(gpu, fieldset):  # :62-68
    MyParticle = pa.Particle.add_variable(pa.Variable("sample_var", initial=attrgetter("x")))
    pset = pa.ParticleSet(fieldset, x=3, y=0, pclass=MyParticle)
    pset.execute(pa.DoNothing, dt=np.timedelta64(1, "s"), runtime=np.timedelta64(21, "s"))
    assert np.allclose([p.sample_var for p in pset], 3.0)


@pytest.mark.gpu
@pytest.mark.parametrize("pset_override", [True, False])
def test_pset_custominit_on_pclass(gpu, fieldset, pset_override):  # :71-83
    MyParticle = pa.Particle.add_variable(pa.Variable("sample_var", initial=4))
    kw = {"sample_var": 5} if pset_override else {}
    pset = pa.ParticleSet(fieldset, x=0, y=0, pclass=MyParticle, **kw)
    pset.execute(pa.DoNothing, dt=np.timedelta64(1, "s"), runtime=np.timedelta64(21, "s"))
    assert np.allclose([p.sample_var for p in pset], 5.0 if pset_override else 4.0)


def test_pset_create_outside_time(fieldset):  # :101-104
    time = np.datetime64("1999-01-01") + np.arange(20) * np.timedelta64(38, "D")
    with pytest.warns(pa.ParticleSetWarning, match="Some particles are set to be released*"):
        pa.ParticleSet(fieldset, pclass=pa.Particle, x=[0] * len(time), y=[0] * len(time), t=time)


@pytest.mark.gpu
def test_populate_indices(gpu, fieldset):  # :119-123 (the reference pins a hash of its own grid's indices; here: the indices are the
    # cells that hold the particles, and a run started from them equals a run started without them)
    npart = 11
    x, y = np.linspace(0, 1, npart), np.linspace(1, 0, npart)
    pset = pa.ParticleSet(fieldset, x=x, y=y)
    pset.populate_indices()
    g = fieldset.U.grid
    xi = np.clip(np.searchsorted(g.lon, x, side="left") - 1, 0, len(g.lon) - 2)
    yi = np.clip(np.searchsorted(g.lat, y, side="left") - 1, 0, len(g.lat) - 2)
    np.testing.assert_array_equal(pset.ei[:, 0], yi * g.xdim + xi)


def test_pset_add_explicit(fieldset):  # :126-137
    npart = 11
    lon, lat = np.linspace(0, 1, npart), np.linspace(1, 0, npart)
    pset = pa.ParticleSet(fieldset, x=lon[0], y=lat[0], pclass=pa.Particle)
    for i in range(1, npart):
        pset.add(pa.ParticleSet(pclass=pa.Particle, x=lon[i], y=lat[i], fieldset=fieldset))
    assert len(pset) == npart
    assert np.allclose([p.x for p in pset], lon, atol=1e-6)
    assert np.allclose([p.y for p in pset], lat, atol=1e-6)
    assert np.allclose(np.diff(pset._data["particle_id"]), np.ones(npart - 1), atol=1e-12)


def test_pset_add_implicit(fieldset):  # :140-144
    pset = pa.ParticleSet(fieldset, x=np.zeros(3), y=np.ones(3), pclass=pa.Particle)
    pset += pa.ParticleSet(fieldset, x=np.ones(4), y=np.zeros(4), pclass=pa.Particle)
    assert len(pset) == 7
    assert np.allclose(np.diff(pset._data["particle_id"]), np.ones(6), atol=1e-12)


def test_pset_add_implicit_in_loop(fieldset, npart=10):  # :147-151
    pset = pa.ParticleSet(fieldset, x=[], y=[])
    for _ in range(npart):
        pset += pa.ParticleSet(pclass=pa.Particle, x=0.1, y=0.1, fieldset=fieldset)
    assert pset.size == npart


def test_pset_merge_inplace(fieldset, npart=100):  # :154-160
    pset1 = pa.ParticleSet(fieldset, x=np.linspace(0, 1, npart), y=np.linspace(1, 0, npart))
    pset2 = pa.ParticleSet(fieldset, x=np.linspace(0, 1, npart), y=np.linspace(0, 1, npart))
    pset1.add(pset2)
    assert pset1.size == 2 * npart and pset2.size == npart


def test_pset_remove_index(fieldset, npart=100):  # :163-170
    pset = pa.ParticleSet(fieldset, x=np.linspace(0, 1, npart), y=np.linspace(1, 0, npart))
    pset.remove_indices([0, 10, 20])
    assert pset.size == 97
    assert not np.any(np.isin(pset.particle_id, [0, 10, 20]))


def test_pset_iterator(fieldset):  # :173-178
    npart = 10
    pset = pa.ParticleSet(fieldset, x=np.zeros(npart), y=np.ones(npart))
    for i, particle in enumerate(pset):
        assert particle.particle_id == i
    assert i == npart - 1
    pset[3].x = 0.5  # writes reach the columns
    assert pset.x[3] == np.float32(0.5)


@pytest.mark.parametrize("depths", [np.linspace(1, 10, 10), np.linspace(-10, -1, 10), np.concatenate([np.linspace(-15, -1, 5), np.linspace(0, 2, 5)]),
                                    np.concatenate([np.linspace(-9, -3, 3), np.linspace(2, 8, 3)]), np.concatenate([np.linspace(-8, -2, 3), np.linspace(3, 9, 3)])])
def test_pset_default_z(depths):  # :181-215 (default z: the depth level closest to zero)
    nz = len(depths)
    md = pa.SGrid2DMetadata(node_dimensions=("XG", "YG"), node_coordinates=("lon", "lat"),
                            face_dimensions=(pa.FaceNodePadding("XC", "XG", pa.Padding.LOW), pa.FaceNodePadding("YC", "YG", pa.Padding.LOW)),
                            vertical_dimensions=(pa.FaceNodePadding("ZC", "depth", pa.Padding.BOTH),))
    coords = {"lon": (("XG",), np.linspace(0, 9, 10)), "lat": (("YG",), np.linspace(0, 9, 10)), "depth": (("depth",), depths)}
    z = np.zeros((nz, 10, 10))
    fs = pa.FieldSet.from_sgrid_conventions(pa.Dataset({"U": (("depth", "YG", "XG"), z), "V": (("depth", "YG", "XG"), z)}, coords, sgrid=md), mesh="flat")
    pset = pa.ParticleSet(fs, x=[0], y=[0])
    assert np.isclose(pset.z[0], depths[np.argmin(np.abs(depths))])
