"""User-written kernels on the device (the reference's plug-in point #1: ``def kernel(particles, fieldset)``, kernel.py:67-70, run by
the loop of kernel.py:206-216).

A Python function cannot run inside a HIP kernel, but the kernels users write for Parcels are almost always a handful of ELEMENTWISE
statements over particle Variables -- ``particles.age += particles.dt``, ``particles.state = np.where(..., StatusCode.Delete, ...)``,
``particles[particles.state >= 50].state = StatusCode.Delete``, ``particles.temp = fieldset.T[particles]``.  For that class this
module translates the function's AST into one more kernel of the fused step loop (a stage machine like the built-in ones: the
statements up to a field sample, the sample, the statements after it), compiles the kernel-list interpreter with it for the one program
variant the FieldSet needs (``hipcc``, ~5 s, cached on disk by content hash) and registers the module with the library
(``pk_set_user_program``); the kernel list then runs as ONE launch, built-in and user kernels alike, the particle columns never leave
the GPU.  Anything outside the class (``NotTranslatable``: control flow, reductions, ``len(particles)``, random numbers, transcendental
functions, more than PK_MAX_EXTRA touched Variables, ...) keeps running through ``hostkernels.py`` -- the
reference's loop on the host columns.

NumPy's semantics are reproduced statically, per expression: every sub-expression carries its NumPy dtype (NEP 50: Python scalars
are weak), an operation is computed in ``np.result_type`` of its operands (``float32 + float32`` is a float add, ``int32 + float32``
a float64 add), ``/`` of integers is a float64 division, an in-place operator computes in the promoted dtype and casts back to the
column's, an assignment casts like ``ndarray.__setitem__``.  tests/test_gpu_jit_kernels.py compares every supported construct with the
host path (which IS NumPy) bit for bit.

SELECTIONS of the particles -- ``near = particles[particles.d2s < 0.5]``, ``near.dx += u * near.dt``, ``sel.v[mask] = ...``,
``particles[np.where(cond)]`` -- stay elementwise: every array carries the selection it lives on (``_V.dom``: the code of a boolean
slot), arrays combine only on ONE selection (NumPy: equal shapes), a store through a selection is a masked store, and a field sample for
a selection is a conditional request: a lane outside it goes straight on, which is why such a kernel keeps its own stage counter
(``UserKernelSource.counter``).  Samples may be taken at computed points (``fieldset.UV[t, z, y1, x1, particles]``) and, on rectilinear
grids, without the particles (``fieldset.T[t, z, y, x]``: state and ``ei`` are restored around the request).  ``particles.state`` is a
write-through proxy that is read when an operation consumes it (a sample later in the statement may change it); tests on the whole set
(``if len(inds) == 0: return``, ``if np.any(mask): ...``) are dropped when everything they guard stays inside that selection.  What the
reference's loop decides for the whole SET stays out: a kernel that stores ``StatusCode.Success`` (kernel.py:190-193 keeps such
particles running while any other particle is).  tools/survey_user_kernels.py: 66 of the 86 kernels of the reference's own tests and
tutorials compile.
"""

from __future__ import annotations

import ast
import ctypes as C
import hashlib
import inspect
import os
import subprocess
import textwrap

import numpy as np

from . import _hip
from .statuscodes import StatusCode

__all__ = ["NotTranslatable", "translate", "UserProgram", "jit_enabled", "compile_kernel_list"]

PK_KERNEL_USER0, PK_MAX_USER_KERNELS = 40, 8
_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_INCLUDE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")


class NotTranslatable(Exception):
    """The function is outside the elementwise class: it runs on the host path instead."""


def jit_enabled() -> bool:
    return os.environ.get("PARCELS_AMD_JIT", "1") not in ("0", "false", "no")


_CT = {"b": "bool", "i32": "int32_t", "i64": "int64_t", "f32": "float", "f64": "double"}
_NP = {"b": np.bool_, "i32": np.int32, "i64": np.int64, "f32": np.float32, "f64": np.float64}
_WEAK = ("wb", "wi", "wf")


def _ty_of_dtype(dt) -> str:
    dt = np.dtype(dt)
    for k, v in _NP.items():
        if dt == np.dtype(v):
            return k
    raise NotTranslatable(f"dtype {dt} has no device form")


class _V:
    """A typed elementwise expression: C++ code, NumPy dtype tag (or weak Python scalar tag), constant value, array-ness."""

    __slots__ = ("code", "ty", "const", "array", "dom", "explicit", "values", "lazy")

    def __init__(self, code, ty, const=None, array=False, dom=None, explicit=False, values=None, lazy=False):
        self.code, self.ty, self.const, self.array = code, ty, const, array
        # lazy: a bare `particles.state` -- a write-through proxy on the host, read when an OPERATION consumes it (expr() materialises the
        # operation's result there and then: a field sample later in the statement may change the state, field.py:307-378)
        self.lazy = lazy
        # values: the set of values the expression can take where that is known statically (constants, `particles.state` itself -- the
        # string "state" --, np.where over such): what a kernel may store into `state` is checked on it (store_var)
        self.values = values if values is not None else ({const} if const is not None and not isinstance(const, float) else None)
        # dom: the sub-selection of the particles an array is defined on -- None: all of them; otherwise the code of the boolean slot that
        # selects them (`particles[mask].x`, `view.x`, `column[mask]`).  NumPy needs equal shapes to combine arrays: equal domains here.
        self.dom, self.explicit = dom, explicit


class _View:
    """`particles`, `particles[mask]`, `view[mask2]`, `particles[np.where(cond)]`: a selection of the particles; mask None = all."""

    __slots__ = ("mask",)

    def __init__(self, mask):
        self.mask = mask

    @property
    def dom(self):
        return self.mask.code if self.mask is not None else None


def _lit(v, ty) -> str:
    if ty in ("b", "wb"):
        return "true" if v else "false"
    if ty in ("i32", "i64", "wi"):
        return f"{int(v)}LL" if ty != "i32" else f"{int(v)}"
    v = float(v)
    if v != v:
        return "__builtin_nan(\"\")"
    if v in (float("inf"), float("-inf")):
        return ("-" if v < 0 else "") + "__builtin_inf()"
    s = repr(v)
    if "e" not in s and "." not in s and "n" not in s:
        s += ".0"
    return f"({s})"


def _const(v) -> _V:
    if isinstance(v, (bool, np.bool_)):
        return _V(_lit(v, "wb"), "wb" if isinstance(v, bool) else "b", bool(v))
    if isinstance(v, np.generic):
        ty = _ty_of_dtype(v.dtype)
        return _V(f"(({_CT[ty]}){_lit(v.item(), 'wf' if ty[0] == 'f' else 'wi')})", ty, v.item())
    if isinstance(v, int):  # IntEnum (StatusCode) included
        return _V(_lit(int(v), "wi"), "wi", int(v))
    if isinstance(v, float):
        return _V(_lit(v, "wf"), "wf", float(v))
    raise NotTranslatable(f"constant of type {type(v).__name__}")


def _np_arg(v: _V):
    if v.ty == "wb":
        return True
    if v.ty == "wi":
        return int(v.const) if v.const is not None and abs(int(v.const)) < 2**31 else 1
    if v.ty == "wf":
        return 1.5
    return np.dtype(_NP[v.ty])


def _promote(a: _V, b: _V) -> str:
    if a.ty in _WEAK and b.ty in _WEAK:
        return "wf" if "wf" in (a.ty, b.ty) else ("wi" if "wi" in (a.ty, b.ty) else "wb")
    return _ty_of_dtype(np.result_type(_np_arg(a), _np_arg(b)))


def _cast(v: _V, ty: str) -> str:
    if ty in _WEAK:
        return v.code
    if v.ty == ty:
        return v.code
    return f"(({_CT[ty]})({v.code}))"


def _strong(ty: str) -> str:
    """The dtype a weak Python scalar takes when it becomes an array element by itself."""
    return {"wb": "b", "wi": "i64", "wf": "f64"}.get(ty, ty)


_SPATIAL = ("x", "y", "z", "dx", "dy", "dz")
_NEXT_STAGE = "/*next-stage*/ "  # where a request is made: case_body writes the stage counter of kernels with conditional samples here
# numpy name -> (C function, arity) of the transcendental functions accepted under PARCELS_AMD_JIT_LIBM=1
_LIBM = {"sin": ("sin", 1), "cos": ("cos", 1), "tan": ("tan", 1), "arcsin": ("asin", 1), "arccos": ("acos", 1), "arctan": ("atan", 1),
         "arctan2": ("atan2", 2), "exp": ("exp", 1), "log": ("log", 1), "log10": ("log10", 1), "sinh": ("sinh", 1), "cosh": ("cosh", 1),
         "tanh": ("tanh", 1), "hypot": ("hypot", 2)}


class _Translator(ast.NodeVisitor):
    def __init__(self, func, pclass, fieldset, var_slot, field_ids, next_dt_f32=False, slot_prefix=""):
        self.func, self.fieldset, self.field_ids = func, fieldset, field_ids
        self.slot_prefix = slot_prefix
        self.var_slot = var_slot  # user Variable name -> (extra column index, dtype tag)
        self.vars = {v.name: np.dtype(v.dtype) for v in pclass.variables}
        self.spatial = _ty_of_dtype(self.vars["x"])
        self.next_dt_f32 = next_dt_f32
        src = textwrap.dedent(inspect.getsource(func))
        tree = ast.parse(src)
        fdef = tree.body[0]
        if not isinstance(fdef, ast.FunctionDef) or len(fdef.args.args) != 2 or fdef.args.vararg or fdef.args.kwarg or fdef.decorator_list:
            raise NotTranslatable("not a plain `def kernel(particles, fieldset)`")
        self.pname, self.fname = fdef.args.args[0].arg, fdef.args.args[1].arg
        self.fdef = fdef
        self.env = dict(func.__globals__)
        if func.__closure__:
            for name, cell in zip(func.__code__.co_freevars, func.__closure__):
                try:
                    self.env[name] = cell.cell_contents
                except ValueError:
                    pass
        self.locals: dict[str, _V] = {}
        self.decl: list[str] = []  # members of PkUserLocals
        self.stages: list[list[str]] = [[]]
        self.touched: set[str] = set()
        self.sampled: list = []  # ('UV' | 'UVW' | scalar field id) of every sample, in order
        self.views: dict[str, _View] = {}  # locals bound to a selection of the particles
        self.index_locals: dict[str, _V] = {}  # locals bound to np.where(cond) / np.flatnonzero(cond): usable as particles[<local>] only
        self._kids: list[list[_V]] = []  # operands of the expression being translated (their domains combine into the result's)
        self._stmt_masks: dict[str, _V] = {}  # mask expression -> its slot, within the statement being translated
        self._combined: dict[tuple, _V] = {}  # (view domain, mask code) -> the slot of their conjunction
        self.shared_temps: set[str] = set()  # local arrays bound to a second name (`w = u`): ONE ndarray on the host
        self.confined: list[_V] = []  # selections whose emptiness an `if` tested: stores and samples must stay inside them
        self.conditional = False  # some sample is taken for a sub-selection: the stage machine keeps its own counter (case_body)
        self.detached = False  # some sample is taken without the particles (state and `ei` untouched)
        self.aliases: set[str] = set()  # locals bound to a bare `particles.<var>`: a write-through view on the host, not a temporary
        self.nslot = 0

    # ---- helpers -------------------------------------------------------------------------------------------------------------
    def emit(self, line):
        self.stages[-1].append(line)

    def new_slot(self, ty, prefix="l") -> str:
        name = f"{self.slot_prefix}{prefix}{self.nslot}"
        self.nslot += 1
        self.decl.append(f"{_CT[ty]} {name};")
        return f"L.ul.{name}"

    def var_type(self, name) -> str:
        if name in _SPATIAL:
            return self.spatial
        if name in ("t", "dt"):
            return "f64"
        if name == "next_dt":
            if self.next_dt_f32 or np.dtype(self.vars.get("next_dt", np.float64)) != np.float64:
                raise NotTranslatable("a float32 next_dt Variable")
            return "f64"
        if name == "state":
            return "i32"
        if name == "particle_id":
            return "i64"
        if name in self.var_slot:
            return self.var_slot[name][1]
        raise NotTranslatable(f"particle Variable '{name}' has no device column")

    def load_var(self, name) -> _V:
        if name not in self.vars:
            raise NotTranslatable(f"particles have no Variable '{name}'")
        ty = self.var_type(name)
        self.touched.add(name)
        if name in _SPATIAL or name in ("t", "dt", "next_dt"):
            code = f"p.{name}" if ty == "f64" else f"((float)p.{name})"
        elif name == "state":
            return _V("((int32_t)c.state)", ty, array=True, values={"state"}, lazy=True)
        elif name == "particle_id":
            code = "((int64_t)p.id)"
        else:
            code = f"(({_CT[ty]}*)a.p.extra[{self.var_slot[name][0]}])[c.row]"
        return _V(code, ty, array=True)

    def store_var(self, name, value: _V, mask: _V | None = None):
        if name not in self.vars:
            raise NotTranslatable(f"particles have no Variable '{name}'")
        if name in ("particle_id", "ei"):
            raise NotTranslatable(f"assignment to particles.{name}")
        ty = self.var_type(name)
        self.touched.add(name)
        self.check_confined(mask, f"a store into particles.{name}")
        val = _cast(value, ty)
        if name in _SPATIAL or name in ("t", "dt", "next_dt"):
            stmt = f"p.{name} = (double)({val});"
        elif name == "state":
            # kernel.py:190-193: a particle in state Success is still evaluated, for as long as ANY particle of the set is in state Evaluate -- a
            # property of the whole set at every iteration, which lanes that run their step loops independently cannot know
            if value.values is None:
                raise NotTranslatable("a computed value stored into particles.state (only status codes, particles.state and np.where over them)")
            if int(StatusCode.Success) in {int(v) for v in value.values if v != "state"}:
                raise NotTranslatable("StatusCode.Success stored into particles.state: the reference's loop keeps such particles running while any "
                                      "other particle is (kernel.py:190-193) -- the host path does that")
            stmt = f"c.state = (int)({val});"
        else:
            stmt = f"(({_CT[ty]}*)a.p.extra[{self.var_slot[name][0]}])[c.row] = {val};"
        self.emit(f"if ({mask.code}) {{ {stmt} }}" if mask is not None else stmt)

    def try_const(self, node):
        """Evaluate a sub-tree that involves neither the particles nor a field nor a local: module constants, np.pi, StatusCode.X,
        fieldset.<context constant>."""
        for n in ast.walk(node):
            if isinstance(n, ast.Name) and (n.id == self.pname or n.id in self.locals):
                return None
            if isinstance(n, (ast.Call, ast.Subscript, ast.Lambda, ast.ListComp, ast.GeneratorExp)):
                return None
        env = dict(self.env)

        class _Ctx:
            pass

        ctx = _Ctx()
        for k, v in dict(self.fieldset.context).items():
            setattr(ctx, k, v)
        env[self.fname] = ctx
        try:
            val = eval(compile(ast.fix_missing_locations(ast.Expression(body=node)), "<kernel>", "eval"), env)  # noqa: S307 -- the user's own kernel source
        except Exception:
            return None
        if isinstance(val, (bool, int, float, np.generic)):
            return _const(val)
        return None

    # ---- expressions ---------------------------------------------------------------------------------------------------------
    def expr(self, node) -> _V:
        c = self.try_const(node)
        if c is not None:
            return c
        m = getattr(self, "e_" + type(node).__name__, None)
        if m is None:
            raise NotTranslatable(f"expression {type(node).__name__}")
        self._kids.append([])
        try:
            v = m(node)
        finally:
            kids = self._kids.pop()
        if kids and not v.explicit:
            v.dom = self.common_domain(kids)
        if any(k.lazy for k in kids):  # an operation on the state column: it reads the column here (code is emitted where it is used)
            ty = _strong(v.ty)
            slot = self.new_slot(ty, "st")
            self.emit(f"{slot} = {_cast(v, ty)};")
            v = _V(slot, ty, const=None, array=v.array, dom=v.dom, explicit=v.explicit, values=v.values)
        if self._kids:
            self._kids[-1].append(v)
        return v

    @staticmethod
    def common_domain(vs):
        """Arrays combine only on one selection of the particles (NumPy: equal shapes); scalars go with anything."""
        arrays = [v for v in vs if v.array]
        doms = {v.dom for v in arrays}
        if len(doms) > 1:
            raise NotTranslatable("arrays over different selections of the particles in one expression (NumPy would need equal shapes)")
        return doms.pop() if doms else None

    # ---- selections of the particles -----------------------------------------------------------------------------------------------
    def mask_slot(self, m: _V) -> _V:
        """A boolean array as a slot (kept as it is when it already is one: a named mask gives ONE domain wherever it is used)."""
        if m.ty not in ("b", "wb") or not m.array:
            raise NotTranslatable("a selection that is not a boolean mask over the particles")
        if m.code.startswith("L.ul.") and m.code.replace("L.ul.", "").replace("_", "").isalnum() and m.ty == "b":
            return m
        if m.code in self._stmt_masks:  # the same mask expression again within ONE statement (nothing is stored in between): one domain
            return self._stmt_masks[m.code]
        t = self.new_slot("b")
        self.emit(f"{t} = {m.code};")
        self._stmt_masks[m.code] = _V(t, "b", array=True, dom=m.dom, explicit=True)
        return self._stmt_masks[m.code]

    def select(self, view: _View, m: _V) -> _View:
        """view[m]: m is a mask over `view`."""
        if m.dom != view.dom:
            raise NotTranslatable("a mask over another selection of the particles than the one it indexes")
        m = self.mask_slot(m)
        if view.mask is None:
            return _View(_V(m.code, "b", array=True, dom=None, explicit=True))
        key = (view.mask.code, m.code)
        if key not in self._combined:
            t = self.new_slot("b")
            self.emit(f"{t} = {view.mask.code} && {m.code};")
            self._combined[key] = _V(t, "b", array=True, dom=None, explicit=True)
        return _View(self._combined[key])

    def index_mask(self, node):
        """np.where(cond) / np.nonzero(cond) / np.flatnonzero(cond) / np.argwhere(cond).flatten(): the rows where cond holds, in order --
        as a selection of the particles the same thing as the mask itself.  -> the mask or None."""
        if isinstance(node, ast.Name) and node.id in self.index_locals:
            return self.index_locals[node.id]
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr in ("flatten", "ravel") and not node.args \
                and isinstance(node.func.value, ast.Call) and self.np_func(node.func.value) == "argwhere" and len(node.func.value.args) == 1:
            return self.mask_slot(self.expr(node.func.value.args[0]))
        if isinstance(node, ast.Call) and self.np_func(node) in ("where", "nonzero", "flatnonzero") and len(node.args) == 1 and not node.keywords:
            return self.mask_slot(self.expr(node.args[0]))
        return None

    def view_of(self, node):
        """The selection an expression denotes, or None when it is not one."""
        if isinstance(node, ast.Name):
            if node.id == self.pname:
                return _View(None)
            return self.views.get(node.id)
        if isinstance(node, ast.Subscript):
            base = self.view_of(node.value)
            if base is None:
                return None
            m = self.index_mask(node.slice)
            if m is None:
                depth = len(self._kids)
                self._kids.append([])
                try:
                    m = self.expr(node.slice)
                finally:
                    del self._kids[depth:]
            return self.select(base, m)
        return None

    def e_Constant(self, node):
        return _const(node.value)

    def e_Name(self, node):
        if node.id in self.locals:
            return self.locals[node.id]
        if node.id in self.views or node.id in self.index_locals:
            raise NotTranslatable(f"'{node.id}' (a selection of the particles) used as a value")
        raise NotTranslatable(f"name '{node.id}'")

    def e_Attribute(self, node):
        view = self.view_of(node.value)
        if view is not None:
            v = self.load_var(node.attr)
            v.dom, v.explicit = view.dom, True
            return v
        raise NotTranslatable(f"attribute .{node.attr}")

    def is_sample(self, node):
        """`fieldset.F[particles]`, `fieldset.F[t, z, y, x, particles]` or `fieldset.F[t, z, y, x]` (field.py:187-195, 297-304); `particles`
        may be any selection of them (`particles[mask]`, a local bound to one)."""
        if not (isinstance(node, ast.Subscript) and isinstance(node.value, ast.Attribute) and isinstance(node.value.value, ast.Name)
                and node.value.value.id == self.fname):
            return False
        sl = node.slice
        if isinstance(sl, ast.Tuple):
            return len(sl.elts) == 4 or (len(sl.elts) == 5 and self.is_selection(sl.elts[4]))
        return self.is_selection(sl)

    def is_selection(self, node):
        while isinstance(node, ast.Subscript):
            node = node.value
        return isinstance(node, ast.Name) and (node.id == self.pname or node.id in self.views)

    def sample_point(self, node, fld):
        """-> (code of t, z, y, x as doubles, whether y is a float32 array, attached to the particles, mask of the selection or None)."""
        sl = node.slice
        if not isinstance(sl, ast.Tuple):
            view = self.view_of(sl)
            return ("p.t", "p.z", "p.y", "p.x"), "c.pf", True, view.mask
        attached = len(sl.elts) == 5
        view = self.view_of(sl.elts[4]) if attached else None
        depth = len(self._kids)
        self._kids.append([])
        try:
            pts = [self.expr(e) for e in sl.elts[:4]]  # (evaluated left to right, like the subscript tuple)
        finally:
            del self._kids[depth:]
        for v, what in zip(pts, "tzyx"):
            if not v.array or v.ty in ("b", "wb"):
                raise NotTranslatable(f"sample coordinate {what} that is not a numeric array over the particles")
        dom = self.common_domain(pts)
        if attached and dom != view.dom:
            raise NotTranslatable("sample coordinates over another selection of the particles than the one passed along")
        mask = view.mask if attached else (None if dom is None else _V(dom, "b", array=True))
        if not attached:
            # field.py:173-176: no guess from `ei`, no state update, `ei` stays.  On a rectilinear grid the search does not depend on the
            # guess; a curvilinear search that starts from the hash returns float32-rounded cell coordinates (index_search.py:242-295)
            ufld = getattr(fld, "U", None)
            curv = getattr(getattr(ufld if ufld is not None else fld, "grid", None), "is_curvilinear", None)
            if curv is None:
                raise NotTranslatable("sample without particles on a field whose grid is not known here")
            if curv:
                raise NotTranslatable("sample without particles on a curvilinear grid (the search starts from the hash, not from `ei`)")
        # the velocity conversion on a spherical mesh is a float32 cosine when y is a float32 array (_xinterpolators.py:183-187)
        return tuple(f"(double)({v.code})" for v in pts), ("true" if pts[2].ty == "f32" else "false"), attached, mask

    def sample(self, node):
        """`fieldset.F[...]` -> a stage boundary; returns the tuple of sampled components (float64 arrays on the selection sampled)."""
        name = node.value.attr
        fld = self.fieldset.fields.get(name)
        if fld is None:
            raise NotTranslatable(f"fieldset has no field '{name}'")
        vector = hasattr(fld, "U")
        if vector:
            if name not in ("UV", "UVW"):
                raise NotTranslatable(f"vector field '{name}'")
            if fld is not self.fieldset.fields.get(name):
                raise NotTranslatable("vector field alias")
            kind, n, fid = ("RQ_UVW", 3, 0) if name == "UVW" else ("RQ_UV", 2, 0)
        else:
            if name in ("U", "V", "W"):
                raise NotTranslatable("sampling a velocity component by itself (the reference warns: host path)")
            kind, n, fid = "RQ_SCALAR", 1, self.field_ids[name]
        (pt, pz, py, px), yf32, attached, mask = self.sample_point(node, fld)
        if attached:  # (a sample without the particles changes nothing: it may be taken anywhere)
            self.check_confined(mask, f"a sample of fieldset.{name}")
        self.sampled.append(name if vector else int(fid))
        saved = None
        if not attached:
            self.detached = True
            saved = [self.new_slot("i32", "k") for _ in range(5)]
            self.emit(f"{saved[0]} = c.state; {saved[1]} = c.ei0; {saved[2]} = c.ei1; {saved[3]} = c.ei2; {saved[4]} = c.ei3;")
        request = f"rq.kind = {kind}; rq.fidx = {fid}; rq.f32 = {yf32}; rq.t = {pt}; rq.z = {pz}; rq.y = {py}; rq.x = {px}; return false;"
        if mask is not None:  # only the selected particles sample (the others would take the sample's error codes): they go straight on
            self.conditional = True
            request = f"if ({mask.code}) {{ {request} }}"
        self.emit(_NEXT_STAGE + request)
        self.stages.append([])
        self._stmt_masks = {}  # (a mask expression evaluated again behind the sample may see other states: not the same selection)
        if saved:
            # (a sample outside the field's time interval stops the reference with a RuntimeError when no particles came along,
            # field.py:31-37: here the particle keeps the error code, which stops the run as well)
            self.emit(f"c.state = c.state == {int(StatusCode.ErrorOutsideTimeInterval)} ? c.state : {saved[0]}; c.ei0 = {saved[1]}; c.ei1 = {saved[2]}; "
                      f"c.ei2 = {saved[3]}; c.ei3 = {saved[4]};")
        out = []
        dom = mask.code if mask is not None else None
        for j in range(n):
            slot = self.new_slot("f64", "s")
            self.emit(f"{slot} = L.r[{3 + j}];")
            out.append(_V(slot, "f64", array=True, dom=dom, explicit=True))
        return out if vector else out[0]

    def e_Subscript(self, node):
        if self.is_sample(node):
            r = self.sample(node)
            if isinstance(r, list):
                raise NotTranslatable("a vector sample must be unpacked: u, v = fieldset.UV[particles]")
            return r
        if self.view_of(node.value) is None:  # column[mask] / array[mask]: the array on the sub-selection
            depth = len(self._kids)
            self._kids.append([])
            try:
                arr = self.expr(node.value)
                m = self.index_mask(node.slice) or self.expr(node.slice)
            finally:
                del self._kids[depth:]
            if arr.array and m.array and m.ty in ("b", "wb"):
                sub = self.select(_View(None if arr.dom is None else _V(arr.dom, "b", array=True)), m)
                return _V(arr.code, arr.ty, array=True, dom=sub.dom, explicit=True)
        raise NotTranslatable("subscript in an expression")

    def e_UnaryOp(self, node):
        v = self.expr(node.operand)
        if isinstance(node.op, ast.USub):
            if v.ty in ("b", "wb"):
                raise NotTranslatable("negating a boolean")
            return _V(f"(-({v.code}))", v.ty, array=v.array)
        if isinstance(node.op, ast.UAdd):
            return v
        if isinstance(node.op, ast.Invert) and v.ty in ("b", "wb"):
            return _V(f"(!({v.code}))", "b", array=v.array)
        raise NotTranslatable(f"unary {type(node.op).__name__}")

    def e_BinOp(self, node):
        op = node.op
        if isinstance(op, ast.Pow):
            e = self.try_const(node.right)
            if e is None or e.const != 2 or e.ty not in ("wi", "wf"):
                raise NotTranslatable("** with an exponent other than the literal 2")
            v = self.expr(node.left)
            if v.ty in ("b", "wb"):
                raise NotTranslatable("boolean ** 2")
            ty = _strong(v.ty) if (e.ty == "wi" or _strong(v.ty)[0] == "f") else "f64"  # ndarray.__pow__(2) is np.square: the array's dtype
            tmp = self.new_slot(ty)
            self.emit(f"{tmp} = {_cast(v, ty)};")
            return _V(f"({tmp} * {tmp})", ty, array=v.array)
        a, b = self.expr(node.left), self.expr(node.right)
        arr = a.array or b.array
        if isinstance(op, (ast.BitAnd, ast.BitOr, ast.BitXor)):
            if a.ty not in ("b", "wb") or b.ty not in ("b", "wb"):
                raise NotTranslatable("bitwise operator on non-boolean operands")
            sym = {"BitAnd": "&&", "BitOr": "||", "BitXor": "!="}[type(op).__name__]
            return _V(f"(({a.code}) {sym} ({b.code}))", "b", array=arr)
        ty = _promote(a, b)
        if ty in ("b", "wb"):
            raise NotTranslatable("arithmetic on booleans")
        if isinstance(op, (ast.Add, ast.Sub, ast.Mult)):
            sym = {"Add": "+", "Sub": "-", "Mult": "*"}[type(op).__name__]
            return _V(f"({_cast(a, ty)} {sym} {_cast(b, ty)})", ty, array=arr)
        if isinstance(op, ast.Div):  # np.true_divide: integers divide as float64
            if ty in ("i32", "i64"):
                ty = "f64"
            elif ty == "wi":
                ty = "wf"
            ca, cb = (_cast(a, ty), _cast(b, ty)) if ty not in _WEAK else (f"((double)({a.code}))", f"((double)({b.code}))")
            return _V(f"({ca} / {cb})", ty, array=arr)
        if isinstance(op, ast.Mod):  # np.remainder on floats: fmod, then the sign of the divisor (npy_divmod)
            rty = ty
            if ty == "wf":  # two Python floats: float.__mod__ is the same rule in double precision
                ty = "f64"
            if ty not in ("f32", "f64"):
                raise NotTranslatable("% on integers")
            fm = "fmodf" if ty == "f32" else "fmod"
            ta, tb, tm = self.new_slot(ty), self.new_slot(ty), self.new_slot(ty)
            self.emit(f"{ta} = {_cast(a, ty) if a.ty not in _WEAK else f'(({_CT[ty]})({a.code}))'}; {tb} = {_cast(b, ty) if b.ty not in _WEAK else f'(({_CT[ty]})({b.code}))'}; {tm} = {fm}({ta}, {tb});")
            self.emit(f"if ({tb} != 0) {{ if ({tm} != 0) {{ if (({tb} < 0) != ({tm} < 0)) {tm} += {tb}; }} else {{ {tm} = __builtin_copysign{'f' if ty == 'f32' else ''}(({_CT[ty]})0, {tb}); }} }}")
            return _V(tm, rty if rty == "wf" else ty, array=arr)
        raise NotTranslatable(f"operator {type(op).__name__}")

    def e_Compare(self, node):
        if len(node.ops) != 1:
            raise NotTranslatable("chained comparison")
        a, b = self.expr(node.left), self.expr(node.comparators[0])
        sym = {"Lt": "<", "LtE": "<=", "Gt": ">", "GtE": ">=", "Eq": "==", "NotEq": "!="}.get(type(node.ops[0]).__name__)
        if sym is None:
            raise NotTranslatable(f"comparison {type(node.ops[0]).__name__}")
        ty = _promote(a, b)
        ct = _strong(ty)
        return _V(f"({_cast(a, ct)} {sym} {_cast(b, ct)})", "b", array=a.array or b.array)

    def shape_source(self, node):
        """`<array>.shape`, `<array>.size`, `len(<array>)`, also wrapped in a 1-tuple: the array whose shape is meant, or None."""
        if isinstance(node, ast.Tuple) and len(node.elts) == 1:
            node = node.elts[0]
        if isinstance(node, ast.Attribute) and node.attr in ("shape", "size"):
            inner = node.value
        elif isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == "len" and "len" not in self.env and len(node.args) == 1:
            inner = node.args[0]
        else:
            return None
        if self.view_of(inner) is not None:
            return None  # (len() of a selection is not its number of rows in the reference: particlesetview.py:83-84)
        depth = len(self._kids)
        self._kids.append([])
        try:
            v = self.expr(inner)
        finally:
            del self._kids[depth:]
        return v if v.array else None

    def np_func(self, node):
        f = node.func
        if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and self.env.get(f.value.id) is np:
            return f.attr
        if isinstance(f, ast.Name) and f.id == "abs" and "abs" not in self.env:
            return "abs"
        return None

    def e_Call(self, node):
        name = self.np_func(node)
        if name is None or node.keywords:
            raise NotTranslatable("call of something other than a supported numpy function")
        # the ufuncs behind the operators, called by name
        binop = {"add": ast.Add, "subtract": ast.Sub, "multiply": ast.Mult, "divide": ast.Div, "true_divide": ast.Div, "mod": ast.Mod, "remainder": ast.Mod}
        cmpop = {"less": ast.Lt, "less_equal": ast.LtE, "greater": ast.Gt, "greater_equal": ast.GtE, "equal": ast.Eq, "not_equal": ast.NotEq}
        if name in binop and len(node.args) == 2:
            return self.e_BinOp(ast.BinOp(left=node.args[0], op=binop[name](), right=node.args[1]))
        if name in cmpop and len(node.args) == 2:
            return self.e_Compare(ast.Compare(left=node.args[0], ops=[cmpop[name]()], comparators=[node.args[1]]))
        if name == "negative" and len(node.args) == 1:
            return self.e_UnaryOp(ast.UnaryOp(op=ast.USub(), operand=node.args[0]))
        if name == "square" and len(node.args) == 1:
            return self.e_BinOp(ast.BinOp(left=node.args[0], op=ast.Pow(), right=ast.Constant(2)))
        if name in ("zeros", "ones", "full") and len(node.args) == (2 if name == "full" else 1):  # np.zeros(particles.x.shape), np.zeros(len(..))
            like = self.shape_source(node.args[0])
            if like is None:
                raise NotTranslatable(f"np.{name} of a shape other than that of an array over the particles")
            if name == "full":
                fillv = self.expr(node.args[1])
                if fillv.array:
                    raise NotTranslatable("np.full with an array fill value")
                ty = _strong(fillv.ty)
                return _V(_cast(fillv, ty), ty, array=True, dom=like.dom, explicit=True)
            return _V("0.0" if name == "zeros" else "1.0", "f64", array=True, dom=like.dom, explicit=True)
        if name == "isin" and len(node.args) == 2:  # membership in a constant list (e.g. a list of status codes): an OR of equalities
            try:
                test = np.asarray(eval(compile(ast.fix_missing_locations(ast.Expression(body=node.args[1])), "<kernel>", "eval"), dict(self.env)))  # noqa: S307
            except Exception:
                raise NotTranslatable("np.isin with test elements that are not constants of the module") from None
            if test.ndim != 1 or test.size == 0 or test.size > 16 or test.dtype.kind not in "biuf":
                raise NotTranslatable("np.isin with test elements other than a short list of numbers")
            v = self.expr(node.args[0])
            tty = _ty_of_dtype(test.dtype if test.dtype.kind != "u" else np.int64)
            ct = _strong(_promote(v, _V("0", tty)))
            t = self.new_slot(ct)
            self.emit(f"{t} = {_cast(v, ct)};")
            alts = " || ".join(f"({t} == {_cast(_V(_lit(e.item(), 'wf' if tty[0] == 'f' else ('wb' if tty == 'b' else 'wi')), tty), ct)})" for e in test)
            return _V(f"({alts})", "b", array=v.array)
        args = [self.expr(a) for a in node.args]
        arr = any(a.array for a in args)
        if name == "sign" and len(args) == 1:  # -1 / 0 / +1 in the argument's dtype, NaN stays NaN
            v = args[0]
            ty = _strong(v.ty)
            if ty == "b":
                raise NotTranslatable("np.sign of a boolean")
            t = self.new_slot(ty)
            self.emit(f"{t} = {_cast(v, ty)};")
            one = f"(({_CT[ty]})1)"
            nanpart = f"({t} != {t}) ? {t} : " if ty[0] == "f" else ""
            return _V(f"({nanpart}(({t} > 0) ? {one} : (({t} < 0) ? -{one} : (({_CT[ty]})0))))", ty, array=arr)
        if name == "where" and len(args) == 3:
            c, a, b = args
            ty = _strong(_promote(a, b))
            cond = c.code if c.ty in ("b", "wb") else f"(({c.code}) != 0)"
            return _V(f"(({cond}) ? {_cast(a, ty)} : {_cast(b, ty)})", ty, array=arr,
                      values=(a.values | b.values) if a.values is not None and b.values is not None else None)
        if name in ("abs", "absolute", "fabs") and len(args) == 1:
            v = args[0]
            ty = _strong(v.ty)
            if ty == "b":
                raise NotTranslatable("abs of a boolean")
            if ty[0] == "f":
                return _V(f"{'fabsf' if ty == 'f32' else 'fabs'}({_cast(v, ty)})", ty, array=arr)
            t = self.new_slot(ty)
            self.emit(f"{t} = {_cast(v, ty)};")
            return _V(f"({t} < 0 ? -{t} : {t})", ty, array=arr)
        if name in ("sqrt", "floor", "ceil", "trunc") and len(args) == 1:
            v = args[0]
            ty = _strong(v.ty)
            if ty[0] != "f":
                ty = "f64"
            return _V(f"{name}{'f' if ty == 'f32' else ''}({_cast(v, ty)})", ty, array=arr)
        if name == "fmod" and len(args) == 2:
            ty = _strong(_promote(*args))
            if ty[0] != "f":
                raise NotTranslatable("np.fmod on integers")
            return _V(f"{'fmodf' if ty == 'f32' else 'fmod'}({_cast(args[0], ty)}, {_cast(args[1], ty)})", ty, array=arr)
        if name in ("minimum", "maximum") and len(args) == 2:  # NaN-propagating, like the ufuncs
            ty = _strong(_promote(*args))
            if ty == "b":
                raise NotTranslatable(f"np.{name} on booleans")
            ta, tb = self.new_slot(ty), self.new_slot(ty)
            self.emit(f"{ta} = {_cast(args[0], ty)}; {tb} = {_cast(args[1], ty)};")
            # NumPy's loops (SIMD min / max with NaN propagation) return the SECOND operand when both compare equal: maximum(-0.0, 0.0) is
            # 0.0, maximum(0.0, -0.0) is -0.0 -- visible as the sign of a later division by it (found by tests/test_jit_translator_fuzz.py)
            cmp_ = "<" if name == "minimum" else ">"
            nan = f" || {ta} != {ta}" if ty[0] == "f" else ""
            return _V(f"(({ta} {cmp_} {tb}{nan}) ? {ta} : {tb})", ty, array=arr)
        if name == "clip" and len(args) == 3:  # np.clip(a, lo, hi) == np.minimum(np.maximum(a, lo), hi) (the ufunc's definition)
            inner = ast.Call(func=ast.Attribute(value=node.func.value, attr="maximum", ctx=ast.Load()), args=[node.args[0], node.args[1]], keywords=[])
            outer = ast.Call(func=ast.Attribute(value=node.func.value, attr="minimum", ctx=ast.Load()), args=[inner, node.args[2]], keywords=[])
            return self.e_Call(outer)
        if name in ("logical_and", "logical_or") and len(args) == 2:
            ca, cb = (a.code if a.ty in ("b", "wb") else f"(({a.code}) != 0)" for a in args)
            return _V(f"(({ca}) {'&&' if name == 'logical_and' else '||'} ({cb}))", "b", array=arr)
        if name == "logical_not" and len(args) == 1:
            a = args[0]
            return _V(f"(!({a.code if a.ty in ('b', 'wb') else f'(({a.code}) != 0)'}))", "b", array=arr)
        if name in ("isnan", "isfinite") and len(args) == 1:
            v = args[0]
            ty = _strong(v.ty)
            if ty[0] != "f":
                return _V("false" if name == "isnan" else "true", "b", array=arr)
            t = self.new_slot(ty)
            self.emit(f"{t} = {_cast(v, ty)};")
            return _V(f"({t} != {t})" if name == "isnan" else f"(({t} - {t}) == 0)", "b", array=arr)
        if name in ("deg2rad", "radians", "rad2deg", "degrees") and len(args) == 1:  # x * (pi / 180) in the loop's dtype, like the ufunc
            v = args[0]
            ty = _strong(v.ty)
            if ty[0] != "f":
                ty = "f64"
            k = "3.14159265358979323846 / 180.0" if name in ("deg2rad", "radians") else "180.0 / 3.14159265358979323846"
            kk = f"(({_CT[ty]})({k}))" if ty == "f64" else (f"(3.14159265358979323846f / 180.0f)" if name in ("deg2rad", "radians") else "(180.0f / 3.14159265358979323846f)")
            return _V(f"({_cast(v, ty)} * {kk})", ty, array=arr)
        if name in _LIBM and len(args) == _LIBM[name][1]:
            # transcendental functions are NOT bit-identical between NumPy's loops and the device library (each within ~1 ulp of the
            # exact value): only on request -- PARCELS_AMD_JIT_LIBM=1 -- otherwise the kernel keeps NumPy's own values on the host path
            if os.environ.get("PARCELS_AMD_JIT_LIBM", "0") in ("0", "", "false", "no"):
                raise NotTranslatable(f"np.{name} (set PARCELS_AMD_JIT_LIBM=1 to accept device transcendentals: within ~1 ulp of NumPy's, not bit-identical)")
            ty = _strong(args[0].ty) if len(args) == 1 else _strong(_promote(*args))
            if ty[0] != "f":
                ty = "f64"
            fn = _LIBM[name][0] + ("f" if ty == "f32" else "")
            return _V(f"{fn}({', '.join(_cast(a, ty) for a in args)})", ty, array=arr)
        if name in ("float32", "float64", "int32", "int64") and len(args) == 1:
            ty = {"float32": "f32", "float64": "f64", "int32": "i32", "int64": "i64"}[name]
            return _V(f"(({_CT[ty]})({args[0].code}))", ty, array=arr)
        if name in ("zeros_like", "ones_like", "empty_like") and len(args) == 1 or name == "full_like" and len(args) == 2:
            like = args[0]  # a new array of the argument's dtype on the argument's selection
            if not like.array:
                raise NotTranslatable(f"np.{name} of a scalar")
            ty = _strong(like.ty)
            fill = {"zeros_like": "0", "ones_like": "1", "empty_like": "0"}.get(name)
            if fill is None:
                if args[1].array:
                    raise NotTranslatable("np.full_like with an array fill value")
                fill = f"({args[1].code})"
            return _V(f"(({_CT[ty]})({fill}))", ty, array=True, dom=like.dom, explicit=True)
        raise NotTranslatable(f"np.{name}")

    # ---- statements ----------------------------------------------------------------------------------------------------------
    def target(self, node):
        """-> ('local', name) | ('var', name, mask or None): mask = the selection of the particles the store goes to (its code is the
        domain an array value must have)."""
        if isinstance(node, ast.Name):
            if node.id in (self.pname, self.fname):
                raise NotTranslatable("rebinding a kernel argument")
            return ("local", node.id)
        if isinstance(node, ast.Attribute):  # particles.v, particles[mask].v, view.v
            view = self.view_of(node.value)
            if view is not None:
                return ("var", node.attr, view.mask)
        if isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name) and node.value.id in self.locals:  # temporary[mask]
            cur = self.locals[node.value.id]
            if not cur.array or node.value.id in self.aliases or node.value.id in self.shared_temps:
                raise NotTranslatable("item assignment on a local that is a scalar, a particle column's view or one array under two names")
            m = self.index_mask(node.slice)
            if m is None:
                depth = len(self._kids)
                self._kids.append([])
                try:
                    m = self.expr(node.slice)
                finally:
                    del self._kids[depth:]
            sub = self.select(_View(None if cur.dom is None else _V(cur.dom, "b", array=True)), m)
            return ("item", node.value.id, sub.mask)
        if isinstance(node, ast.Subscript) and isinstance(node.value, ast.Attribute):  # particles.v[mask], view.v[mask]
            view = self.view_of(node.value.value)
            if view is not None:
                m = self.index_mask(node.slice)
                if m is None:
                    depth = len(self._kids)
                    self._kids.append([])
                    try:
                        m = self.expr(node.slice)
                    finally:
                        del self._kids[depth:]
                return ("var", node.value.attr, self.select(view, m).mask)
        raise NotTranslatable("assignment target")

    def emptiness_test(self, test):
        """`len(S) == 0`, `not len(S)`, `S.size == 0` -> (S, True); `len(S) > 0`, `len(S)`, `np.any(mask)`, `mask.any()` -> (S, False), S the mask
        of the selection (None: all particles); anything else -> None."""
        def selection(node):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == "len" and "len" not in self.env and len(node.args) == 1:
                node = node.args[0]
            elif isinstance(node, ast.Attribute) and node.attr == "size":
                node = node.value
            else:
                return False
            if isinstance(node, ast.Name) and node.id in self.index_locals:
                return self.index_locals[node.id]
            view = self.view_of(node.value if isinstance(node, ast.Attribute) and self.view_of(node.value) is not None else node)
            return view.mask if view is not None else False

        if isinstance(test, ast.Compare) and len(test.ops) == 1 and isinstance(test.comparators[0], ast.Constant):
            sel, k, op = selection(test.left), test.comparators[0].value, type(test.ops[0]).__name__
            if sel is not False and ((op == "Eq" and k == 0) or (op == "Lt" and k == 1) or (op == "LtE" and k == 0)):
                return sel, True
            if sel is not False and ((op == "Gt" and k == 0) or (op == "GtE" and k == 1) or (op == "NotEq" and k == 0)):
                return sel, False
            return None
        if isinstance(test, ast.UnaryOp) and isinstance(test.op, ast.Not):
            sel = selection(test.operand)
            return (sel, True) if sel is not False else None
        sel = selection(test)
        if sel is not False:
            return sel, False
        mask_node = None
        if isinstance(test, ast.Call) and self.np_func(test) == "any" and len(test.args) == 1 and not test.keywords:
            mask_node = test.args[0]
        elif isinstance(test, ast.Call) and isinstance(test.func, ast.Attribute) and test.func.attr == "any" and not test.args and not test.keywords:
            mask_node = test.func.value
        if mask_node is not None:
            depth = len(self._kids)
            self._kids.append([])
            try:
                m = self.expr(mask_node)
            finally:
                del self._kids[depth:]
            if m.array and m.ty in ("b", "wb"):
                return self.select(_View(None if m.dom is None else _V(m.dom, "b", array=True)), m).mask, False
        return None

    def within(self, mask, sel) -> bool:
        """Is the selection `mask` (None: all particles) a sub-selection of `sel`?"""
        if mask is None:
            return False
        if mask.code == sel.code:
            return True
        for (parent, _), combined in self._combined.items():
            if combined.code == mask.code:
                return self.within(_V(parent, "b", array=True), sel)
        return False

    def check_confined(self, mask, what):
        for sel in self.confined:
            if not self.within(mask, sel):
                raise NotTranslatable(f"{what} outside the selection whose emptiness an `if` tests (the test then decides for the whole set)")

    @staticmethod
    def check_store_domain(value: _V, mask):
        if value.array and value.dom != (mask.code if mask is not None else None):
            raise NotTranslatable("assignment of an array over another selection of the particles (NumPy would need matching shapes)")

    def assign(self, tgt, value: _V, alias=False):
        if tgt[0] == "local":
            ty = _strong(value.ty)
            if value.array:
                slot = self.new_slot(ty)
                self.emit(f"{slot} = {_cast(value, ty)};")
                self.locals[tgt[1]] = _V(slot, ty, array=True, dom=value.dom, explicit=True)
            else:
                self.locals[tgt[1]] = value  # a scalar stays a (weak) scalar
            (self.aliases.add if alias else self.aliases.discard)(tgt[1])
            self.shared_temps.discard(tgt[1])
            self.views.pop(tgt[1], None)
            self.index_locals.pop(tgt[1], None)
        elif tgt[0] == "item":  # temporary[mask] = value: the temporary with the masked elements replaced (cast to ITS dtype)
            _, lname, mask = tgt
            cur = self.locals[lname]
            self.check_store_domain(value, mask)
            slot = self.new_slot(cur.ty)
            self.emit(f"{slot} = ({mask.code}) ? {_cast(value, cur.ty)} : {cur.code};")
            self.locals[lname] = _V(slot, cur.ty, array=True, dom=cur.dom, explicit=True)
        else:
            _, name, mask = tgt
            self.check_store_domain(value, mask)
            self.store_var(name, value, mask)

    def run(self):
        self.run_body(self.fdef.body, top=True)
        self.emit("return true;")

    def run_body(self, body, top):
        for i, st in enumerate(body):
            self._stmt_masks = {}
            if isinstance(st, ast.Expr) and isinstance(st.value, ast.Constant) and isinstance(st.value.value, str):
                continue
            if isinstance(st, ast.Pass):
                continue
            if isinstance(st, ast.Expr) and self.is_sample(st.value):  # a sample for its effect on the particles' state
                self.sample(st.value)
                continue
            if isinstance(st, ast.Assign):
                if len(st.targets) != 1:
                    raise NotTranslatable("chained assignment")
                t = st.targets[0]
                if isinstance(t, ast.Tuple):
                    if not self.is_sample(st.value):
                        raise NotTranslatable("tuple assignment of something other than a vector sample")
                    comps = self.sample(st.value)
                    if not isinstance(comps, list) or len(comps) != len(t.elts):
                        raise NotTranslatable("vector sample unpacked into the wrong number of names")
                    for el, comp in zip(t.elts, comps):
                        if isinstance(el, ast.Name) and el.id == "_":
                            continue
                        self.assign(self.target(el), comp)
                else:
                    if isinstance(t, ast.Name) and t.id not in (self.pname, self.fname):
                        if isinstance(st.value, ast.Subscript):
                            view = self.view_of(st.value)  # ptcls = particles[mask]
                            if view is not None:
                                self.views[t.id] = view
                                self.locals.pop(t.id, None)
                                self.index_locals.pop(t.id, None)
                                continue
                        im = self.index_mask(st.value) if isinstance(st.value, ast.Call) else None  # inds = np.where(cond)
                        if im is not None:
                            self.index_locals[t.id] = im
                            self.locals.pop(t.id, None)
                            self.views.pop(t.id, None)
                            continue
                    bare = isinstance(st.value, ast.Attribute) and self.view_of(st.value.value) is not None
                    if isinstance(st.value, ast.Name) and st.value.id in self.locals and self.locals[st.value.id].array and isinstance(t, ast.Name):
                        twin = st.value.id  # w = u: two names for one ndarray -- in-place changes through either would show in both
                    else:
                        twin = None
                    value = self.expr(st.value)  # Python's order: the value, then the target's selection (a sample in the value may change states)
                    self.assign(self.target(t), value, alias=bare)
                    if twin is not None:
                        self.shared_temps.update((twin, t.id))
                        if twin in self.aliases:
                            self.aliases.add(t.id)
                continue
            if isinstance(st, ast.If):
                c = self.try_const(st.test)
                if c is not None and not c.array:  # a condition that is a constant of the run (fieldset.<context>, module constants): one branch
                    self.run_body(st.body if c.const else st.orelse, top=False)
                    continue
                # `if len(inds) == 0: return` / `if len(inds) > 0: ...` / `if np.any(mask): ...`: whether a SELECTION is empty is a property of
                # the whole set -- but code that only touches that selection does nothing when it is empty, so the test can go
                emptiness = self.emptiness_test(st.test)
                if emptiness is not None and not st.orelse:
                    sel, when_empty = emptiness
                    only_return = len(st.body) == 1 and isinstance(st.body[0], ast.Return) and st.body[0].value is None
                    if when_empty and only_return and top:
                        if sel is not None:  # (len(particles.x) == 0 never holds: a kernel is called with particles)
                            self.confined.append(sel)  # ... everything from here on must stay inside the selection
                        continue
                    if not when_empty and sel is not None:
                        self.confined.append(sel)
                        try:
                            self.run_body(st.body, top=False)
                        finally:
                            self.confined.pop()
                        continue
                raise NotTranslatable("`if` on something other than a constant of the run (elementwise code has no control flow)")
            if isinstance(st, ast.AugAssign):
                tgt = self.target(st.target)
                if tgt[0] == "local":
                    # a local TEMPORARY (ndarray on the host): in-place ufunc, dtype kept, the result must cast back with 'same_kind'
                    lname = tgt[1]
                    if lname not in self.locals or lname in self.aliases or lname in self.shared_temps or not self.locals[lname].array:
                        raise NotTranslatable("in-place operator on a local that is a particle column's view, a scalar or one array under two names")
                    if not isinstance(st.op, (ast.Add, ast.Sub, ast.Mult, ast.Div)):
                        raise NotTranslatable(f"in-place {type(st.op).__name__}")
                    cur, val = self.locals[lname], self.expr(st.value)
                    if val.array and val.dom != cur.dom:
                        raise NotTranslatable("in-place operator with an array over another selection of the particles")
                    ty = _promote(cur, val)
                    if isinstance(st.op, ast.Div) and ty in ("i32", "i64"):
                        ty = "f64"
                    if not np.can_cast(np.dtype(_NP[_strong(ty)]), np.dtype(_NP[cur.ty]), "same_kind"):
                        raise NotTranslatable(f"in-place operator: {_strong(ty)} does not cast back to {cur.ty} (NumPy raises)")
                    sym = {"Add": "+", "Sub": "-", "Mult": "*", "Div": "/"}[type(st.op).__name__]
                    ct = _strong(ty)
                    slot = self.new_slot(cur.ty)
                    self.emit(f"{slot} = {_cast(_V(f'({_cast(cur, ct)} {sym} {_cast(val, ct)})', ct), cur.ty)};")
                    self.locals[lname] = _V(slot, cur.ty, array=True, dom=cur.dom, explicit=True)
                    continue
                if tgt[0] == "item":  # temporary[mask] += value
                    _, lname, mask = tgt
                    if not isinstance(st.op, (ast.Add, ast.Sub, ast.Mult, ast.Div)):
                        raise NotTranslatable(f"in-place {type(st.op).__name__}")
                    cur, val = self.locals[lname], self.expr(st.value)
                    self.check_store_domain(val, mask)
                    ty = _promote(cur, val)
                    if isinstance(st.op, ast.Div) and ty in ("i32", "i64"):
                        ty = "f64"
                    if not np.can_cast(np.dtype(_NP[_strong(ty)]), np.dtype(_NP[cur.ty]), "same_kind"):
                        raise NotTranslatable(f"in-place operator: {_strong(ty)} does not cast back to {cur.ty} (NumPy raises)")
                    sym = {"Add": "+", "Sub": "-", "Mult": "*", "Div": "/"}[type(st.op).__name__]
                    ct = _strong(ty)
                    slot = self.new_slot(cur.ty)
                    self.emit(f"{slot} = ({mask.code}) ? {_cast(_V(f'({_cast(cur, ct)} {sym} {_cast(val, ct)})', ct), cur.ty)} : {cur.code};")
                    self.locals[lname] = _V(slot, cur.ty, array=True, dom=cur.dom, explicit=True)
                    continue
                _, name, mask = tgt
                cur = self.load_var(name)
                val = self.expr(st.value)
                self.check_store_domain(val, mask)
                if not isinstance(st.op, (ast.Add, ast.Sub, ast.Mult, ast.Div)):
                    raise NotTranslatable(f"in-place {type(st.op).__name__}")
                ty = _promote(cur, val)
                if isinstance(st.op, ast.Div) and ty in ("i32", "i64"):
                    ty = "f64"
                # ufunc(..., out=column): the result must cast back with 'same_kind'
                if not np.can_cast(np.dtype(_NP[_strong(ty)]), np.dtype(_NP[cur.ty]), "same_kind"):
                    raise NotTranslatable(f"in-place operator: {_strong(ty)} does not cast back to {cur.ty} (NumPy raises)")
                sym = {"Add": "+", "Sub": "-", "Mult": "*", "Div": "/"}[type(st.op).__name__]
                ct = _strong(ty)
                self.store_var(name, _V(f"({_cast(cur, ct)} {sym} {_cast(val, ct)})", ct, array=True), mask)
                continue
            if isinstance(st, ast.Return) and st.value is None and top and i == len(body) - 1:
                continue
            raise NotTranslatable(f"statement {type(st).__name__}")


class UserKernelSource:
    def __init__(self, name, decl, stages, touched, sampled=(), detached=False, counter=None):
        self.name, self.decl, self.stages, self.touched, self.sampled = name, decl, stages, touched, list(sampled)
        self.detached = detached  # samples without the particles: restores PCtx::state / ei, which the dedicated kernels keep elsewhere
        # kernels that sample for a sub-selection of the particles: a lane outside it goes straight on to the statements behind the sample,
        # so the stage the caller counts (one per request) is not the place in the kernel any more -- the kernel keeps its own (this slot)
        self.counter = counter

    def case_body(self) -> str:
        out = ["switch (stage) {"] if self.counter is None else [f"if (stage == 0) {self.counter} = 0;", f"switch ({self.counter}) {{"]
        for k, lines in enumerate(self.stages):
            out.append(f"    case {k}: {{")
            mark = "" if self.counter is None else f"{self.counter} = {k + 1}; "
            out += ["        " + ln.replace(_NEXT_STAGE, mark) for ln in lines]
            out.append("    }" if self.counter is None else "    }  // fall through")
        out.append("    default: return true;")
        out.append("}")
        return "\n".join(out)


def translate(func, pclass, fieldset, var_slot, field_ids, next_dt_f32=False, slot_prefix="") -> UserKernelSource:
    """Python kernel -> stage-machine source.  var_slot: {user Variable name: (extra column index, dtype tag 'f32' | 'f64')}."""
    try:
        tr = _Translator(func, pclass, fieldset, var_slot, field_ids, next_dt_f32, slot_prefix)
    except (OSError, TypeError, SyntaxError, IndentationError) as e:
        raise NotTranslatable(f"source of {getattr(func, '__name__', func)!r} is not available: {e}") from None
    tr.run()
    counter = tr.new_slot("i32", "stage") if tr.conditional else None
    return UserKernelSource(func.__name__, tr.decl, tr.stages, tr.touched, tr.sampled, tr.detached, counter)


def candidate_variables(func, pclass):
    """User Variables a kernel mentions (``particles.<name>`` with <name> a non-core Variable): what must be bound as device columns."""
    try:
        tree = ast.parse(textwrap.dedent(inspect.getsource(func)))
    except (OSError, TypeError, SyntaxError, IndentationError):
        return []
    core = set(_SPATIAL) | {"t", "dt", "next_dt", "state", "particle_id", "ei"}
    names = [v.name for v in pclass.variables]
    out = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Attribute) and n.attr in names and n.attr not in core and n.attr not in out:
            out.append(n.attr)
    return out


_TEMPLATE = """// generated by parcels_amd/jit.py -- user kernels: {names}
#define PK_USER_KERNELS 1
#ifndef PK_MIN_WAVES
#define PK_MIN_WAVES 2
#endif
#include <stdint.h>
namespace pk {{
struct PkUserLocals {{
{decl}
}};
}}
#include "pk_kernels.h"
namespace pk {{
PK_DEV bool user_prepare(const KArgs& a, int uk, int stage, int kslot, PCtx& c, PState& p, KLocal& L, Request& rq) {{
    switch (uk) {{
{cases}
        default: c.state = PK_ERROR; return true;
    }}
}}
}}  // namespace pk
extern "C" void pk_user_launch(const void* kargs, int32_t prog, int32_t key, int32_t lds, uint64_t lds_bytes, void* stream) {{
    using namespace pk;
    const KArgs& a = *(const KArgs*)kargs;
    if (prog != 0) {{  // the dedicated A-grid kernel with the user kernels riding along (pk_kernels.h: side_kernel)
{fast_launch}
        fprintf(stderr, "parcels_amd user program: no dedicated kernel (%d, %d) in this module\\n", prog, key);
        abort();
    }}
    if (key != {key} || lds != {lds}) {{
        fprintf(stderr, "parcels_amd user program built for variant (%d, %d), launched as (%d, %d)\\n", {key}, {lds}, key, lds);
        abort();
    }}
    constexpr int WG = wg_size({kind}, {ldsb});
    hipLaunchKernelGGL((advect_kernel<{ft}, {kind}, {interp}, -1, {ldsb}, false>), dim3((unsigned)((a.p.n + WG - 1) / WG)), dim3(WG), (size_t)lds_bytes,
                       (hipStream_t)stream, a);
}}
"""


def _csrc_hash() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(_CSRC)):
        if f.endswith((".h", ".hip")) and f not in ("pk_api.hip",):
            h.update(open(os.path.join(_CSRC, f), "rb").read())
    h.update(open(os.path.join(_INCLUDE, "parcels_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def cache_dir() -> str:
    # (next to the package by default so that modules built in the build container travel to the GPU box with it; the name sorts behind
    # libparcels_hip.so in a listing of the shared objects a process loaded.  A read-only package directory falls back to a per-user one.)
    d = os.environ.get("PARCELS_AMD_JIT_CACHE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "user_kernel_cache")
    try:
        os.makedirs(d, exist_ok=True)
        if not os.access(d, os.W_OK):
            raise PermissionError(d)
    except OSError:
        d = os.path.join(os.path.expanduser("~"), ".cache", "parcels_amd", "user_kernel_cache")
        os.makedirs(d, mode=0o700, exist_ok=True)
    return d


class UserProgram:
    """One compiled module: the kernel-list interpreter of ONE variant with up to PK_MAX_USER_KERNELS user kernels in it."""

    def __init__(self, sources: list[UserKernelSource], key: int, lds: int, fast: int = 0, particles_f32: bool = False):
        """key / lds: the interpreter variant (pk_generic_variant); fast: 1 / 2 = also carry the dedicated A-grid kernel (2-D / 3-D) with
        the user kernels riding along -- only for modules whose kernels sample no field and leave next_dt alone."""
        if not (1 <= len(sources) <= PK_MAX_USER_KERNELS):
            raise NotTranslatable(f"1 .. {PK_MAX_USER_KERNELS} user kernels per kernel list")
        self.sources = list(sources)
        decl = "\n".join("    " + d for s in sources for d in s.decl) or "    char unused;"
        # the locals of different kernels never live at the same time, but they are few: one struct, distinct names
        cases = "\n".join(f"        case {k}: {{\n" + textwrap.indent(s.case_body(), "            ") + "\n        }" for k, s in enumerate(sources))
        ft = "float" if key >= 6 else "double"
        kind, interp = (key % 6) // 3, key % 3
        # what the kernels sample decides whether the list may ride in a dedicated kernel (pk_generic_variant was asked with the same lists)
        sampled = [x for src in sources for x in src.sampled]
        self.sample_fids = sorted({x for x in sampled if isinstance(x, int)})
        self.sample_flags = (2 if "UV" in sampled else 0) | (4 if "UVW" in sampled else 0)  # PK_USER_SAMPLES_UV / _UVW
        rides = fast and all("next_dt" not in src.touched and not src.detached for src in sources) and len(self.sample_fids) <= 4
        self.flags = (1 | self.sample_flags) if rides else self.sample_flags  # PK_USER_RIDE: the module carries the dedicated kernel
        fast_launch = ""
        if self.flags & 1:
            fkey = (2 if key >= 6 else 0) + (1 if particles_f32 else 0)
            d3 = "true" if fast in (2, 4) else "false"
            pfm = 1 if particles_f32 else 0
            if fast in (1, 2):
                launch = (f"hipLaunchKernelGGL((advect_fast_kernel<{ft}, {pfm}, {d3}>), dim3((unsigned)((a.p.n + 255) / 256)), dim3(256), "
                          f"(size_t)lds_bytes, (hipStream_t)stream, a);")
            else:
                # (the two variants of pk_prog_cgrid_fast.hip: FastC::near_edges, the edge cosines of CGrid_Velocity from the sample's own)
                grid_ = "dim3((unsigned)((a.p.n + FC_LANES - 1) / FC_LANES)), dim3(FC_LANES), (size_t)lds_bytes, (hipStream_t)stream, a"
                launch = (f"if (a.fastc.near_edges) hipLaunchKernelGGL((advect_cgrid_kernel<{ft}, {pfm}, {d3}, true>), {grid_}); "
                          f"else hipLaunchKernelGGL((advect_cgrid_kernel<{ft}, {pfm}, {d3}, false>), {grid_});")
            fast_launch = f"        if (prog == {int(fast)} && key == {fkey}) {{\n            {launch}\n            return;\n        }}"
        self.source = _TEMPLATE.format(names=", ".join(s.name for s in sources), decl=decl, cases=cases, key=key, lds=lds, ft=ft, kind=kind,
                                       interp=interp, ldsb="true" if lds else "false", fast_launch=fast_launch)
        # (measured on C2 with a sampling kernel riding in the dedicated A-grid kernel, tools/bench_user_kernels.py: 4 waves per SIMD 12.6 ms,
        # 3 waves 13.8 ms, 2 waves 12.5 ms -- the library's own occupancy target stays; PARCELS_AMD_JIT_FAST_WAVES overrides for A/B runs)
        # A dedicated kernel that carries user kernels keeps the correctly rounded quotients and full-range sines / cosines of the general
        # program (pk_fast_agrid.h: PK_FAST_LEAN, pk_fast_cgrid.h: PK_CG_NEAR / PK_CG_LEAN): a user kernel may branch on a sampled value or a
        # position, and the compiled list has to do what the host path (general program between NumPy kernels) does, bit for bit
        self.defines = ["-DPK_FAST_LEAN=0", "-DPK_CG_NEAR=0", "-DPK_CG_LEAN=0"]
        if os.environ.get("PARCELS_AMD_JIT_FAST_WAVES"):
            self.defines.append("-DPK_MIN_WAVES_FAST=" + os.environ["PARCELS_AMD_JIT_FAST_WAVES"])
        self.digest = hashlib.sha256((self.source + " ".join(self.defines) + _csrc_hash()).encode()).hexdigest()[:20]
        self.path = os.path.join(cache_dir(), f"user_{self.digest}.so")
        self._lib = None

    _failed: dict = {}  # digest -> compiler message: a module that did not compile is not compiled again by every pset.execute

    @classmethod
    def clear_failed(cls):
        """Forget remembered compilation failures (e.g. after fixing the toolchain of a long-running process)."""
        cls._failed.clear()

    def build(self):
        if os.path.exists(self.path):
            return self.path
        if self.digest in UserProgram._failed:
            raise RuntimeError(UserProgram._failed[self.digest])
        # several ranks of one node compile the same module at the same time: private temporaries, atomic rename
        tag = f"{os.getpid()}"
        src = self.path[:-3] + f".{tag}.hip"
        tmp = self.path + f".{tag}.tmp"
        with open(src, "w") as f:
            f.write(self.source)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-function", *self.defines,
               f"-I{_CSRC}", f"-I{_INCLUDE}", src, "-o", tmp]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            msg = f"hipcc failed on the generated user-kernel module {src}:\n{r.stderr[-4000:]}"
            # only a verdict of the COMPILER on this source is remembered; a transient failure (no space left in the cache directory, a
            # signal, hipcc itself missing -> OSError above) is tried again by the next pset.execute
            if r.returncode > 0 and "error:" in r.stderr and "No space left" not in r.stderr:
                UserProgram._failed[self.digest] = msg
            raise RuntimeError(msg)
        os.replace(src, self.path[:-3] + ".hip")  # (kept next to the module: what was compiled)
        os.replace(tmp, self.path)
        return self.path

    def launcher(self) -> int:
        """Address of pk_user_launch (loads the module; its code object registers with the HIP runtime of this process)."""
        if self._lib is None:
            self._lib = C.CDLL(self.build())
        return C.cast(self._lib.pk_user_launch, C.c_void_p).value


def compile_kernel_list(functions, builtin_id, pclass, fieldset, engine, samples=None, device_variables=()):
    """A kernel list with Python functions in it -> (kernel ids, UserProgram, device Variable names), or NotTranslatable.

    functions: the list as the user wrote it; builtin_id(f) -> PK_KERNEL_* id or None for a Python function to translate; pclass: the
    particle class (``.variables`` with ``.name`` / ``.dtype``); fieldset: the parcels_amd.FieldSet on ``engine``; samples /
    device_variables: what SampleField tokens of the list already claimed."""
    funcs = [f for f in functions if builtin_id(f) is None]
    if not funcs:
        raise NotTranslatable("no Python function in the list")
    if len(funcs) > PK_MAX_USER_KERNELS:
        raise NotTranslatable(f"more than {PK_MAX_USER_KERNELS} Python kernels in one list")
    names = {v.name: v for v in pclass.variables}
    dev_vars = list(device_variables)
    for f in funcs:
        for vn in candidate_variables(f, pclass):
            if np.dtype(names[vn].dtype) not in (np.dtype(np.float32), np.dtype(np.float64), np.dtype(np.int32), np.dtype(np.int64)):
                raise NotTranslatable(f"Variable '{vn}' is {np.dtype(names[vn].dtype)}: device columns are float32 / float64 / int32 / int64")
            if vn not in dev_vars:
                dev_vars.append(vn)
    if len(dev_vars) > _hip.PK_MAX_EXTRA:
        raise NotTranslatable(f"more than {_hip.PK_MAX_EXTRA} user Variables touched by device kernels (PK_MAX_EXTRA)")
    var_slot = {vn: (k, _ty_of_dtype(names[vn].dtype)) for k, vn in enumerate(dev_vars)}
    next_dt_f32 = "next_dt" in names and np.dtype(names["next_dt"].dtype) != np.float64
    sources, ids, j = [], [], 0
    for f in functions:
        kid = builtin_id(f)
        if kid is None:
            sources.append(translate(f, pclass, fieldset, var_slot, engine.field_ids, next_dt_f32, slot_prefix=f"k{j}_"))
            kid = PK_KERNEL_USER0 + j
            j += 1
        ids.append(kid)
    # which programs this list runs as on this FieldSet: the variant of the kernel-list interpreter, and whether the dedicated A-grid kernel
    # would take it if the user kernels sample nothing
    prm = engine.make_params(ids, endtime=0.0, dt0=1.0, context=fieldset.context, samples=samples or {})
    sampled = [x for src in sources for x in src.sampled]
    fids = sorted({x for x in sampled if isinstance(x, int)})
    sflags = (2 if "UV" in sampled else 0) | (4 if "UVW" in sampled else 0)
    key, lds, typed, fast = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    cf = (C.c_int32 * 4)(*(fids + [0] * 4)[:4])
    engine.ctx.check(engine.lib.pk_generic_variant(engine.ctx.handle, C.byref(prm), sflags, min(len(fids), 4), cf, C.byref(key), C.byref(lds), C.byref(typed),
                                                   C.byref(fast)), "pk_generic_variant")
    if typed.value:
        raise NotTranslatable("float32 coordinate arrays (NumPy dtype propagation of the typed program)")
    prog = UserProgram(sources, key.value, lds.value, fast=(fast.value if len(fids) <= 4 else 0), particles_f32=np.dtype(names["x"].dtype) == np.float32)
    prog.launcher()  # builds (or finds in the cache) and loads the module
    return ids, prog, dev_vars


_ = (StatusCode, _hip)  # (re-exported names user kernels commonly reference; keeps linters quiet)
