"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz by running the REFERENCE'S OWN hot path.

Run in the build container (where /root/reference exists):

    python -m oracle.make_golden            # all cases
    python -m oracle.make_golden NAME ...   # selected cases

For every case in ``oracle/cases.py`` this wires the reference's real ``XGrid``/``Field``/``VectorField``/
``ParticleSet``/``Kernel`` objects (oracle/ref_shim.py), calls the reference's ``ParticleSet.execute`` and stores
inputs + the resulting particle SoA dict.  The fixtures are what travels to the GPU box; ``/root/reference`` does not.

Stochastic kernels: the reference draws from NumPy's global MT19937 stream (``_advectiondiffusion.py:37-38``), which
cannot be matched on a GPU.  For the golden run ``np.random.normal`` is replaced by the framework's counter-based
generator (Philox4x32-10 keyed by seed/kernel slot, counter = particle_id and time bits; ``philox_normal_pair``
below == ``po_normal_pair`` in parcels_oracle.c) so that the *arithmetic* of the kernels is compared exactly.
"""

from __future__ import annotations

import json
import os
import sys

import numpy as np

from . import cases as cases_mod
from . import ref_shim as rs

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ---- counter-based normal pairs (NumPy restatement of po_normal_pair) -----------------------------------------------
def philox_normal_pair(seed: int, kslot: int, particle_id: np.ndarray, t: np.ndarray):
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    pid = np.asarray(particle_id).astype(np.int64).view(np.uint64)
    tb = np.ascontiguousarray(np.asarray(t, dtype=np.float64)).view(np.uint64)
    c0, c1 = pid & mask, pid >> np.uint64(32)
    c2, c3 = tb & mask, tb >> np.uint64(32)
    k0 = np.uint64(((seed & 0xFFFFFFFF) ^ ((0x9E3779B9 * (kslot + 1)) & 0xFFFFFFFF)) & 0xFFFFFFFF)
    k1 = np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n1 = p1 & mask
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        n3 = p0 & mask
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    a = (c1 << np.uint64(32)) | c0
    b = (c3 << np.uint64(32)) | c2
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    u2 = ((b >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    r = np.sqrt(-2.0 * np.log(u1))
    th = 6.283185307179586476925286766559 * u2
    return r * np.cos(th), r * np.sin(th)


class _NormalPatch:
    """Replaces np.random.normal while a wrapped stochastic kernel runs."""

    def __init__(self, seed):
        self.seed = seed
        self.cur = None
        self.calls = 0

    def begin(self, kslot, particles):
        self.cur = (kslot, np.asarray(particles.particle_id).copy(), np.asarray(particles.t).copy())
        self.calls = 0

    def normal(self, loc, scale):
        kslot, pid, t = self.cur
        z0, z1 = philox_normal_pair(self.seed, kslot, pid, t)
        z = z0 if self.calls == 0 else z1
        self.calls += 1
        return loc + scale * z


def _recovery_kernels(m):
    SC = m["statuscodes"].StatusCode

    def DeleteParticle(particles, fieldset):  # tests/common_kernels.py:12-13
        particles.state = np.where(particles.state >= 50, SC.Delete, particles.state)

    def DeleteOutOfBounds(particles, fieldset):  # tests/test_advection.py:157-161
        particles.state = np.where(particles.state == SC.ErrorOutOfBounds, SC.Delete, particles.state)
        particles.state = np.where(particles.state == SC.ErrorThroughSurface, SC.Delete, particles.state)

    def SubmergeParticle(particles, fieldset):  # tests/test_advection.py:163-174
        if len(particles.state) == 0:
            return
        inds = np.argwhere(particles.state == SC.ErrorThroughSurface).flatten()
        if len(inds) == 0:
            return
        (u, v) = fieldset.UV[particles[inds]]
        particles[inds].dx = u * particles[inds].dt
        particles[inds].dy = v * particles[inds].dt
        particles[inds].dz = 0.0
        particles[inds].z = 0
        particles[inds].state = SC.Evaluate

    def DoNothing(particles, fieldset):  # tests/common_kernels.py:8-9
        pass

    def MoveEast(particles, fieldset):  # tests/common_kernels.py:16-17
        particles.dx += 0.1

    def MoveNorth(particles, fieldset):  # tests/common_kernels.py:20-21
        particles.dy += 0.1

    return {"DeleteParticle": DeleteParticle, "DeleteOutOfBounds": DeleteOutOfBounds, "SubmergeParticle": SubmergeParticle,
            "DoNothing": DoNothing, "MoveEast": MoveEast, "MoveNorth": MoveNorth}


def build_ref_fieldset(case):
    sizes_extra = {}
    for name, dims in case["field_dims"].items():
        for d, s in zip(dims, np.asarray(case["fields"][name]).shape):
            if d in ("XC", "YC", "ZC"):
                sizes_extra[d] = s
    g = rs.make_ref_grid(lon=case["lon"], lat=case["lat"], depth=case.get("depth"), mesh=case["mesh"],
                         x_pad=case.get("x_pad", "low"), y_pad=case.get("y_pad", "low"), z_pad=case.get("z_pad", "both"),
                         sizes_extra=sizes_extra)
    fields = {n: (np.asarray(a), tuple(case["field_dims"][n])) for n, a in case["fields"].items()}
    fs = rs.make_ref_fieldset(grid=g, fields=fields, time_s=case.get("time_s"), cgrid=bool(case.get("cgrid")),
                              constants=case.get("constants") or None, const_mesh=case.get("const_mesh", "flat"),
                              slip=case.get("slip"))
    for k, v in (case.get("context") or {}).items():
        fs.add_context(k, v)
    return fs, g


def ref_sample_case(case):
    """Field.eval(t, z, y, x) of the reference with the requested scalar interpolator (field.py:145-195)."""
    m = rs.load_reference()
    fs, g = build_ref_fieldset(case)
    name = case["sample_field"]
    f = fs.fields[name]
    f.interp_method = getattr(m["xinterp"], case["scalar_interp"][name])()
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        val = f.eval(np.asarray(case["t0"], dtype=float), np.asarray(case["z"], dtype=float), np.asarray(case["y"], dtype=float),
                     np.asarray(case["x"], dtype=float))
    return {"value": np.asarray(val, dtype=np.float64)}, None, {}


def ref_run_case(case):
    """Run one case through the reference. Returns (soa_dict, error_name, extras)."""
    if case.get("kind") == "sample":
        return ref_sample_case(case)
    m = rs.load_reference()
    K = m["kernels"]
    fs, g = build_ref_fieldset(case)
    rec = _recovery_kernels(m)
    patch = _NormalPatch(int(case.get("seed", 0)))
    klist = []
    samples = case.get("sample_into") or {}
    for slot, name in enumerate(case["kernels"]):
        if name in rec:
            klist.append(rec[name])
            continue
        if name in samples:  # the user kernel of the tutorials, in Python, run by the reference's own kernel loop (kernel.py:206-216)
            fname, vname, _ = samples[name]

            def make_sample(fname=fname, vname=vname, name=name):
                def sample(particles, fieldset):
                    if isinstance(vname, (list, tuple)):  # particles.u, particles.v[, particles.w] = fieldset.UV[W][particles]
                        for vn, val in zip(vname, getattr(fieldset, fname)[particles]):
                            if vn is not None:
                                setattr(particles, vn, val)
                    else:
                        setattr(particles, vname, getattr(fieldset, fname)[particles])
                sample.__name__ = name
                return sample
            klist.append(make_sample())
            continue
        f = getattr(K, name)
        if name in ("AdvectionDiffusionM1", "AdvectionDiffusionEM", "DiffusionUniformKh"):
            def make(f=f, slot=slot):
                def wrapped(particles, fieldset):
                    patch.begin(slot, particles)
                    return f(particles, fieldset)
                wrapped.__name__ = f.__name__
                return wrapped
            klist.append(make())
        else:
            klist.append(f)
    extra_vars = None
    pkw = None
    if "AdvectionRK45" in case["kernels"]:
        extra_vars = [("next_dt", np.dtype(case.get("next_dt_dtype", "float64")).type, float(case.get("next_dt0", case["dt"])))]
    for fname, vname, vdt in samples.values():
        for vn in (vname if isinstance(vname, (list, tuple)) else [vname]):
            if vn is not None:
                extra_vars = (extra_vars or []) + [(vn, np.dtype(vdt).type, 0)]
    n = len(np.atleast_1d(case["x"]))
    z = case.get("z")
    if z is not None and np.ndim(z) == 0:
        z = np.full(n, z)
    old_normal = np.random.normal
    np.random.normal = patch.normal
    try:
        out, err = rs.run_reference(
            fs, klist, x=np.asarray(case["x"]), y=np.asarray(case["y"]), z=z, t=case.get("t0"), dt=float(case["dt"]),
            runtime=case.get("runtime"), endtime_s=case.get("endtime"), spatial_dtype=np.dtype(case.get("spatial_dtype", "float64")).type,
            extra_vars=extra_vars, particle_kwargs=pkw, populate=bool(case.get("populate")), outputdt=case.get("outputdt"),
            more_calls=case.get("more_calls") or (),
        )
    finally:
        np.random.normal = old_normal
    extras = {}
    if g._spatialhash is not None:
        # the table itself is megabytes; the fixture stores its SHA-256 so that the framework's own host build
        # (parcels_amd/spatialhash.py) can be pinned against the reference's table on the GPU box too
        sh = g._spatialhash
        ht = sh._hash_table
        import hashlib

        cs = {}
        for k, dt in (("keys", np.uint32), ("starts", np.int64), ("counts", np.int64), ("faces", np.uint32)):
            a = np.ascontiguousarray(np.asarray(ht[k]).astype(dt))
            cs[k] = hashlib.sha256(a.tobytes()).hexdigest()
            cs["n_" + k] = int(a.size)
        cs["bitwidth"] = int(sh._bitwidth)
        cs["bbox"] = [float(v) for v in (sh._xmin, sh._xmax, sh._ymin, sh._ymax, sh._zmin, sh._zmax)]
        extras["hash_checksum"] = cs
    return out, err, extras


def save_case(path, case, out, err, extras):
    arrs = {}
    meta = {}
    for k, v in case.items():
        if k in ("fields",):
            for fn, a in v.items():
                arrs["field__" + fn] = np.asarray(a)
        elif isinstance(v, np.ndarray):
            arrs["in__" + k] = v
        else:
            meta[k] = v
    meta["field_names"] = list(case["fields"].keys())
    meta["err"] = err
    for k, v in out.items():
        arrs["out__" + k] = np.asarray(v)
    if "hash_checksum" in extras:
        meta["hash_checksum"] = extras["hash_checksum"]
    arrs["meta_json"] = np.frombuffer(json.dumps(meta, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o)).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrs)


def load_case(path):
    """Inverse of save_case: returns (case, out, err)."""
    z = np.load(path, allow_pickle=False)
    meta = json.loads(bytes(z["meta_json"]).decode())
    case = {k: v for k, v in meta.items() if k not in ("field_names", "err")}
    case["fields"] = {fn: z["field__" + fn] for fn in meta["field_names"]}
    for k in z.files:
        if k.startswith("in__"):
            case[k[4:]] = z[k]
    case["field_dims"] = {k: tuple(v) for k, v in case["field_dims"].items()}
    out = {k[5:]: z[k] for k in z.files if k.startswith("out__")}
    return case, out, meta["err"]


def main(argv):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    allc = cases_mod.all_cases()
    names = argv or list(allc)
    for name in names:
        case = allc[name]
        out, err, extras = ref_run_case(case)
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        save_case(path, case, out, err, extras)
        if "state" not in out:
            print(f"{name:36s} sampled {len(out['value'])} points, {int(np.sum(out['value'] == 0))} zeros, size={os.path.getsize(path) // 1024} KB")
            continue
        st = np.bincount(out["state"], minlength=1) if len(out["state"]) else []
        codes = {int(i): int(c) for i, c in enumerate(st) if c}
        print(f"{name:36s} n={len(out['x']):4d} err={err} states={codes} size={os.path.getsize(path) // 1024} KB")


if __name__ == "__main__":
    main(sys.argv[1:])
