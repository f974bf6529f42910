"""CPU: differential fuzzing of the kernel translator (parcels_amd/jit.py).  Every seed writes a random elementwise kernel -- arithmetic
over float32 / float64 / int32 / int64 Variables, Python and NumPy scalars, comparisons, np.where / abs / minimum / maximum / floor / sqrt,
%, ** 2, plain and in-place and masked assignments, local temporaries, scalar and vector field samples (the stage boundaries of the generated
kernel, answered by arrays the test supplies) -- into a module, translates it, compiles the emitted C++ for the HOST
(tests/test_jit_translator.py: the shim of the device structs) and demands the columns NumPy produces from the same Python function, bit
for bit.  What is undefined in C and NumPy alike (float -> integer casts of NaN / out-of-range values, integer overflow) is not generated.
Offline sweeps of 10 000 + 6 000 seeds (PARCELS_JIT_FUZZ_SEEDS=10000, 4 minutes on 8 cores) found one real difference -- np.maximum / np.minimum
return their SECOND operand when both compare equal (the sign of a zero, visible after a division) -- fixed; 7 kernels were refused
(`%` of integer constants), none differed.  A second generator (GenViews) writes kernels over SELECTIONS of the particles -- 12 000 seeds
offline, none differed; run on the reference's own ParticleSetView next to HostParticles it showed that len() of a kernel's particles is the
size of the whole set there."""
import importlib.util
import os

import numpy as np
import pytest

import test_jit_translator as T

FLOATS = ["particles.x", "particles.y", "particles.dx", "particles.dy", "particles.age", "particles.acc", "particles.dt", "fieldset.c1", "fieldset.c2"]
INTS = ["particles.count", "particles.flag", "particles.state"]


class Gen:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.locals_f, self.locals_i = [], []

    def pick(self, xs):
        return xs[int(self.rng.integers(0, len(xs)))]

    def const(self, integer):
        if integer:
            return str(int(self.rng.integers(-3, 5)))
        return self.pick(["0.5", "-1.25", "2.0", "3", "0.1", "1e-3", "-7"])

    def iexpr(self, depth):
        """an expression that stays integer (bounded: products of at most two leaves)"""
        r = self.rng.random()
        if depth <= 0 or r < 0.3:
            return self.pick(INTS + self.locals_i) if self.rng.random() < 0.75 else self.const(True)
        if r < 0.55:
            return f"({self.iexpr(depth - 1)} {self.pick(['+', '-'])} {self.iexpr(depth - 1)})"
        if r < 0.65:
            return f"({self.pick(INTS)} * {self.const(True)})"
        if r < 0.8:
            return f"np.where({self.cond(depth - 1)}, {self.iexpr(depth - 1)}, {self.iexpr(depth - 1)})"
        if r < 0.9:
            return f"np.abs({self.iexpr(depth - 1)})"
        return f"np.{self.pick(['minimum', 'maximum'])}({self.iexpr(depth - 1)}, {self.iexpr(depth - 1)})"

    def fexpr(self, depth):
        r = self.rng.random()
        if depth <= 0 or r < 0.22:
            return self.pick(FLOATS + self.locals_f) if self.rng.random() < 0.8 else self.const(False)
        if r < 0.5:
            a = self.fexpr(depth - 1)
            b = self.fexpr(depth - 1) if self.rng.random() < 0.7 else self.iexpr(depth - 1)
            if self.rng.random() < 0.5:
                a, b = b, a
            op = self.pick(['+', '-', '*', '/'])
            if op == "/" and b.strip("()-").replace(".", "").isdigit() and float(b.strip("()-")) == 0:
                b = "2"  # (a Python constant divided by the constant 0 raises in Python itself)
            return f"({a} {op} {b})"
        if r < 0.58:
            return f"(-{self.fexpr(depth - 1)})"
        if r < 0.66:
            return f"np.where({self.cond(depth - 1)}, {self.fexpr(depth - 1)}, {self.pick([self.fexpr(depth - 1), self.iexpr(depth - 1), self.const(False)])})"
        if r < 0.74:
            return f"np.{self.pick(['minimum', 'maximum'])}({self.fexpr(depth - 1)}, {self.pick([self.fexpr(depth - 1), self.const(False), self.iexpr(depth - 1)])})"
        if r < 0.8:
            return f"np.abs({self.fexpr(depth - 1)})"
        if r < 0.85:
            return f"np.sqrt(np.abs({self.fexpr(depth - 1)}))"
        if r < 0.9:
            return f"({self.fexpr(depth - 1)}) ** 2"
        if r < 0.95:
            return f"({self.fexpr(depth - 1)} % {self.pick(['1.5', '-0.75', 'fieldset.c1', '2'])})"
        return f"({self.iexpr(depth - 1)} / {self.pick(['2', '3', 'particles.count', '0.5'])})"

    def cond(self, depth):
        r = self.rng.random()
        op = self.pick(["<", "<=", ">", ">=", "==", "!="])
        # (the left operand always involves a column: a comparison of two Python constants is a Python bool, whose `~` is an integer)
        fl = f"({self.pick(FLOATS[:7])} {self.pick(['+', '-', '*'])} {self.fexpr(depth)})"
        il = f"({self.pick(INTS)} {self.pick(['+', '-'])} {self.iexpr(depth)})"
        if r < 0.45:
            c = f"({fl} {op} {self.pick([self.fexpr(depth), self.const(False)])})"
        elif r < 0.8:
            c = f"({il} {op} {self.pick([self.iexpr(depth), self.const(True)])})"
        else:
            c = f"np.isnan({fl})"
        if depth > 0 and self.rng.random() < 0.3:
            c = f"({c} {self.pick(['&', '|'])} {self.cond(depth - 1)})"
        if self.rng.random() < 0.15:
            c = f"(~{c})"
        return c

    def statement(self):
        r = self.rng.random()
        fvars = ["age", "acc", "dx", "dy", "dz"]
        ivars = ["count", "flag"]
        if r < 0.3:
            return f"particles.{self.pick(fvars)} = {self.fexpr(3)}"
        if r < 0.45:
            return f"particles.{self.pick(fvars)} {self.pick(['+=', '-=', '*=', '/='])} {self.pick([self.fexpr(2), self.iexpr(2), self.const(False)])}"
        if r < 0.55:
            return f"particles.{self.pick(ivars)} = {self.iexpr(3)}"
        if r < 0.62:
            return f"particles.{self.pick(ivars)} {self.pick(['+=', '-='])} {self.pick([self.iexpr(1), self.const(True)])}"
        if r < 0.72:
            return f"particles[{self.cond(2)}].{self.pick(fvars)} = {self.const(False)}"
        if r < 0.78:
            return f"particles.{self.pick(ivars)}[{self.cond(1)}] = {self.const(True)}"
        if r < 0.84:
            return f"particles.{self.pick(fvars)}[{self.cond(1)}] {self.pick(['+=', '*='])} {self.const(False)}"
        if r < 0.93:
            name = f"f{len(self.locals_f)}"
            st = f"{name} = {self.fexpr(2)} * 1"  # (`* 1`: a temporary, not a view of a column)
            self.locals_f.append(name)
            return st
        name = f"i{len(self.locals_i)}"
        st = f"{name} = {self.iexpr(2)} + 0"
        self.locals_i.append(name)
        return st

    def kernel(self, name):
        """-> (source, [(field name, components), ...] in sampling order)"""
        nst = int(self.rng.integers(3, 8))
        nsam = int(self.rng.choice([0, 0, 1, 2, 3]))
        where = sorted(int(v) for v in self.rng.integers(0, nst + 1, size=nsam))
        body, samples = [], []
        for k in range(nst + 1):
            for _ in range(where.count(k)):  # a field sample: a stage boundary of the kernel; its value is a local from here on
                j = len(samples)
                kind = self.rng.random()
                if kind < 0.6:
                    fname = self.pick(["T", "S"])
                    body.append(f"s{j} = fieldset.{fname}[particles]" if self.rng.random() < 0.6 else f"particles.{self.pick(['age', 'acc', 'dz'])} = fieldset.{fname}[particles] * 2 - s_prev".replace("s_prev", self.pick(self.locals_f + ["particles.age"])))
                    if body[-1].startswith("s"):
                        self.locals_f.append(f"s{j}")
                    samples.append((fname, 1))
                elif kind < 0.85:
                    body.append(f"u{j}, v{j} = fieldset.UV[particles]")
                    self.locals_f += [f"u{j}", f"v{j}"]
                    samples.append(("UV", 2))
                else:
                    body.append(f"_, particles.dy, w{j} = fieldset.UVW[particles]")
                    self.locals_f.append(f"w{j}")
                    samples.append(("UVW", 3))
            if k < nst:
                body.append(self.statement())
        return f"import numpy as np\n\n\ndef {name}(particles, fieldset):\n" + "".join(f"    {s}\n" for s in body), samples


@pytest.mark.parametrize("seed", range(int(os.environ.get("PARCELS_JIT_FUZZ_SEEDS", "48"))))
def test_random_kernel_equals_numpy(tmp_path, seed):
    name = f"K{seed}"
    src, samples = Gen(seed).kernel(name)
    path = tmp_path / f"fuzz_kernel_{seed}.py"
    path.write_text(src)
    spec = importlib.util.spec_from_file_location(f"fuzz_kernel_{seed}", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        T._check(getattr(mod, name), tmp_path, spatial=np.float32 if seed % 2 else np.float64, context={"c1": 0.75, "c2": np.float32(1.5)},
                 seed=1000 + seed, n=256, fields=samples)
    except ZeroDivisionError:
        pytest.skip("the generated kernel divides Python constants by zero: not a kernel")
    except T.jit.NotTranslatable as e:  # refusing is always safe (the kernel then runs on the host path); it must stay rare here
        pytest.skip(f"not translatable: {e}")
    except Exception as e:
        raise AssertionError(f"{type(e).__name__}: {e}\n--- kernel ---\n{src}") from None


def _reference_view():
    from oracle import ref_shim

    if not ref_shim.reference_available():
        return None
    return ref_shim.load_reference()["particlesetview"].ParticleSetView


@pytest.mark.skipif(_reference_view() is None, reason="reference tree not present")
@pytest.mark.parametrize("seed", range(int(os.environ.get("PARCELS_JIT_FUZZ_SEEDS", "48"))))
def test_random_kernel_on_the_references_own_view(tmp_path, seed):
    """The chain is anchored in the reference itself: the same random kernels, run on the REFERENCE's ParticleSetView
    (src/parcels/_core/particlesetview.py, loaded unmodified) and on parcels_amd's HostParticles (what a Python kernel receives on the host
    path, and what the translator is compared with above), leave the same columns, bit for bit."""
    import parcels_amd as pa
    from parcels_amd.hostkernels import HostParticles

    View = _reference_view()
    name = f"K{seed}"
    src, samples = Gen(seed).kernel(name)
    path = tmp_path / f"fuzz_kernel_{seed}.py"
    path.write_text(src)
    spec = importlib.util.spec_from_file_location(f"fuzz_kernel_ref_{seed}", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    func = getattr(mod, name)
    n = 256
    P = pa.get_default_particle(np.float32 if seed % 2 else np.float64).add_variable([
        pa.Variable("age", dtype=np.float32, initial=0), pa.Variable("acc", dtype=np.float64, initial=0),
        pa.Variable("count", dtype=np.int32, initial=0), pa.Variable("flag", dtype=np.int64, initial=0)])
    data = T._columns(P, n, 1000 + seed)
    rng = np.random.default_rng(seed + 100)
    fields = {}
    for fname, ncomp in samples:
        if fname not in fields:
            fields[fname] = T._FakeField([rng.normal(size=n) for _ in range(ncomp)])
    fs = T._FakeFieldSet({"c1": 0.75, "c2": np.float32(1.5)}, fields)
    a = {k: v.copy() for k, v in data.items()}
    b = {k: v.copy() for k, v in data.items()}
    try:
        with np.errstate(all="ignore"):
            try:
                func(View(a, np.ones(n, dtype=bool), P), fs)
            except TypeError as e:  # e.g. `particles.x % 1.5`: the reference's column proxy has no __mod__ (HostParticles is a superset)
                pytest.skip(f"the reference's own view does not support this kernel: {e}")
            func(HostParticles(b, np.arange(n)), fs)
    except ZeroDivisionError:
        pytest.skip("the generated kernel divides Python constants by zero: not a kernel")
    for k in a:
        assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k], equal_nan=True), (k, src)


# ---- selections of the particles ---------------------------------------------------------------------------------------------------------
class GenViews(Gen):
    """Random kernels over SELECTIONS of the particles: locals bound to particles[cond], stores / in-place operators / masked stores through
    them, inline selections, np.where(cond) index selections, samples for a selection (conditional requests: the kernel keeps its own stage
    counter) and samples at computed points."""

    def __init__(self, seed):
        super().__init__(seed)
        self.view_names = []
        self.view_locals = {}  # view name -> float locals that live on it

    def over(self, view, text):
        return text.replace("particles.", f"{view}.")

    def vexpr(self, view, depth, integer=False):
        keep = (self.locals_f, self.locals_i)
        self.locals_f, self.locals_i = list(self.view_locals.get(view, [])), []  # (full-length temporaries do not combine with a selection)
        try:
            e = self.iexpr(depth) if integer else self.fexpr(depth)
        finally:
            self.locals_f, self.locals_i = keep
        return self.over(view, e)

    def vcond(self, view, depth):
        keep = (self.locals_f, self.locals_i)
        self.locals_f, self.locals_i = list(self.view_locals.get(view, [])), []
        try:
            c = self.cond(depth)
        finally:
            self.locals_f, self.locals_i = keep
        return self.over(view, c)

    def view_statement(self):
        fvars, ivars = ["age", "acc", "dx", "dy", "dz"], ["count", "flag"]
        r = self.rng.random()
        if not self.view_names or r < 0.22:
            name = f"sel{len(self.view_names)}"
            if self.view_names and self.rng.random() < 0.3:  # a selection within a selection
                parent = self.pick(self.view_names)
                st = f"{name} = {parent}[{self.vcond(parent, 1)}]"
            elif self.rng.random() < 0.25:
                st = f"{name} = particles[np.where({self.vcond('particles', 1)})]"
            else:
                st = f"{name} = particles[{self.vcond('particles', 1)}]"
            self.view_names.append(name)
            self.view_locals[name] = []
            return st
        v = self.pick(self.view_names)
        if r < 0.45:
            return f"{v}.{self.pick(fvars)} = {self.pick([self.vexpr(v, 2), self.const(False)])}"
        if r < 0.6:
            return f"{v}.{self.pick(fvars)} {self.pick(['+=', '-=', '*='])} {self.pick([self.vexpr(v, 1), self.const(False)])}"
        if r < 0.7:
            return f"{v}.{self.pick(ivars)} = {self.pick([self.vexpr(v, 1, integer=True), self.const(True)])}"
        if r < 0.8:
            return f"{v}.{self.pick(fvars)}[{self.vcond(v, 1)}] = {self.const(False)}"
        if r < 0.88:
            return f"{v}.{self.pick(ivars)}[{self.vcond(v, 1)}] += {self.const(True)}"
        name = f"g{sum(len(x) for x in self.view_locals.values())}_{v}"
        st = f"{name} = {self.vexpr(v, 2)} * 1"
        self.view_locals[v].append(name)
        return st

    def kernel(self, name):
        nst = int(self.rng.integers(4, 10))
        body, samples = [], []
        for k in range(nst):
            r = self.rng.random()
            if r < 0.5:
                body.append(self.view_statement())
            elif r < 0.65 and self.view_names:  # a sample for a selection
                v = self.pick(self.view_names)
                j = len(samples)
                if self.rng.random() < 0.5:
                    fname = self.pick(["T", "S"])
                    key = v if self.rng.random() < 0.5 else f"{v}.t, {v}.z, {v}.y + 0.25, {v}.x - {v}.dx, {v}"
                    body.append(f"q{j}_{v} = fieldset.{fname}[{key}]")
                    self.view_locals[v].append(f"q{j}_{v}")
                    samples.append((fname, 1))
                else:
                    body.append(f"uu{j}_{v}, vv{j}_{v} = fieldset.UV[{v}]")
                    self.view_locals[v] += [f"uu{j}_{v}", f"vv{j}_{v}"]
                    samples.append(("UV", 2))
            elif r < 0.7:  # a block guarded by an emptiness test on a named mask: everything in it stays inside that selection
                m = f"m{k}"
                body.append(f"{m} = {self.vcond('particles', 1)}")
                fvars, ivars = ["age", "acc", "dx", "dy", "dz"], ["count", "flag"]
                inner = [f"particles[{m}].{self.pick(fvars)} = {self.const(False)}",
                         f"particles[{m}].{self.pick(ivars)} += {self.const(True)}",
                         f"particles.{self.pick(fvars)}[{m}] *= {self.const(False)}"]
                if self.rng.random() < 0.5:
                    j = len(samples)
                    fname = self.pick(["T", "S"])
                    inner.append(f"particles[{m}].{self.pick(fvars)} = fieldset.{fname}[particles[{m}]] * 2")
                    samples.append((fname, 1))
                guard = self.pick([f"np.any({m})", f"{m}.any()", f"len(particles[{m}]) > 0"])
                body.append(f"if {guard}:\n" + "\n".join("        " + ln for ln in inner))
            elif r < 0.74:  # a temporary made by an array constructor, filled through masks, used
                b = f"b{k}"
                src_col = self.pick(["particles.x", "particles.age", "particles.acc", "particles.count"])
                ctor = self.pick([f"np.zeros_like({src_col})", f"np.zeros({src_col}.shape)", f"np.full_like({src_col}, {self.const(True)})", f"np.ones(len({src_col}))"])
                body.append(f"{b} = {ctor}")
                body.append(f"{b}[{self.vcond('particles', 1)}] = {self.pick([self.const(True), self.const(False) if 'count' not in src_col or 'like' not in ctor else self.const(True)])}")
                if self.rng.random() < 0.6:
                    body.append(f"{b}[{self.vcond('particles', 1)}] {self.pick(['+=', '-=', '*='])} {self.const(True)}")
                body.append(f"particles.{self.pick(['acc', 'dz', 'age'])} += {b}")
            elif r < 0.78:  # a sample at a computed point, for all particles
                j = len(samples)
                fname = self.pick(["T", "S"])
                body.append(f"p{j} = fieldset.{fname}[particles.t + particles.dt, particles.z, particles.y * 1, particles.x + {self.fexpr(1)}, particles]")
                self.locals_f.append(f"p{j}")
                samples.append((fname, 1))
            else:
                body.append(self.statement())
        return f"import numpy as np\n\n\ndef {name}(particles, fieldset):\n" + "".join(f"    {s}\n" for s in body), samples


def _load(tmp_path, src, name, tag):
    path = tmp_path / f"{tag}.py"
    path.write_text(src)
    spec = importlib.util.spec_from_file_location(tag, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return getattr(mod, name)


@pytest.mark.parametrize("seed", range(int(os.environ.get("PARCELS_JIT_FUZZ_SEEDS", "48"))))
def test_random_kernel_over_selections_equals_numpy(tmp_path, seed):
    name = f"V{seed}"
    src, samples = GenViews(seed).kernel(name)
    func = _load(tmp_path, src, name, f"fuzz_views_{seed}")
    try:
        T._check(func, tmp_path, spatial=np.float32 if seed % 2 else np.float64, context={"c1": 0.75, "c2": np.float32(1.5)},
                 seed=2000 + seed, n=256, fields=samples, check_nsamples=False, check_log=False)
    except ZeroDivisionError:
        pytest.skip("the generated kernel divides Python constants by zero: not a kernel")
    except T.jit.NotTranslatable as e:
        pytest.skip(f"not translatable: {e}")
    except Exception as e:
        raise AssertionError(f"{type(e).__name__}: {e}\n--- kernel ---\n{src}") from None


@pytest.mark.skipif(_reference_view() is None, reason="reference tree not present")
@pytest.mark.parametrize("seed", range(int(os.environ.get("PARCELS_JIT_FUZZ_SEEDS", "48"))))
def test_random_kernel_over_selections_on_the_references_own_view(tmp_path, seed):
    """Selections of selections, masks within them, np.where index selections and samples for a selection: the REFERENCE's ParticleSetView
    and HostParticles leave the same columns (and hand the fields the same points, in the same order)."""
    import parcels_amd as pa
    from parcels_amd.hostkernels import HostParticles

    View = _reference_view()
    name = f"V{seed}"
    src, samples = GenViews(seed).kernel(name)
    func = _load(tmp_path, src, name, f"fuzz_views_ref_{seed}")
    n = 256
    P = pa.get_default_particle(np.float32 if seed % 2 else np.float64).add_variable([
        pa.Variable("age", dtype=np.float32, initial=0), pa.Variable("acc", dtype=np.float64, initial=0),
        pa.Variable("count", dtype=np.int32, initial=0), pa.Variable("flag", dtype=np.int64, initial=0)])
    data = T._columns(P, n, 2000 + seed)
    out, logs = [], []
    for make in (lambda d: View(d, np.ones(n, dtype=bool), P), lambda d: HostParticles(d, np.arange(n), by_mask=True)):  # (as the host loop hands it over)
        rng = np.random.default_rng(seed + 100)
        log, fields = [], {}
        for fname, ncomp in samples:
            if fname not in fields:
                fields[fname] = T._FakeField([rng.normal(size=n) for _ in range(ncomp)], log)
                if ncomp > 1:
                    fields[fname].U = fields[fname].V = None
        fs = T._FakeFieldSet({"c1": 0.75, "c2": np.float32(1.5)}, fields)
        d = {k: v.copy() for k, v in data.items()}
        try:
            with np.errstate(all="ignore"):
                func(make(d), fs)
        except ZeroDivisionError:
            pytest.skip("the generated kernel divides Python constants by zero: not a kernel")
        except TypeError as e:
            if not out:  # e.g. `%`: the reference's column proxy has no __mod__ (HostParticles is a superset)
                pytest.skip(f"the reference's own view does not support this kernel: {e}")
            raise
        out.append(d)
        logs.append(log)
    a, b = out
    for k in a:
        assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k], equal_nan=True), (k, src)
    assert len(logs[0]) == len(logs[1])
    for ea, eb in zip(*logs):
        assert np.array_equal(ea["rows"], eb["rows"]) and ea["f32"] == eb["f32"] and all(np.array_equal(ea[c], eb[c], equal_nan=True) for c in "tzyx"), src
