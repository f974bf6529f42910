"""Which of the user-written kernels in the reference's own tests and tutorials does parcels_amd/jit.py translate, and why not the rest?

Extracts every `def f(particles, fieldset)` from /root/reference (tests/*.py, docs/**/*.ipynb|md), gives it a permissive context
(every `particles.<name>` it mentions exists as a float32 Variable, every `fieldset.<name>[...]` is a scalar field, every other
`fieldset.<name>` a constant) and runs the translator.  Only meaningful where the reference tree is present (the build container).

    python tools/survey_user_kernels.py [--verbose] [--json out.json]
"""
import argparse
import ast
import collections
import glob
import importlib.util
import json
import os
import re
import sys
import tempfile
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("PARCELS_REFERENCE", "/root/reference")


def python_blocks(path):
    if path.endswith(".py"):
        yield open(path).read()
    elif path.endswith(".ipynb"):
        nb = json.load(open(path))
        for cell in nb.get("cells", []):
            if cell.get("cell_type") == "code":
                src = "".join(cell.get("source", []))
                yield "\n".join(ln for ln in src.splitlines() if not ln.lstrip().startswith(("%", "!")))
    else:
        for m in re.finditer(r"```\{?(?:python|code-cell)[^\n]*\n(.*?)```", open(path).read(), re.S):
            yield m.group(1)


def kernel_defs(path):
    for block in python_blocks(path):
        try:
            tree = ast.parse(block)
        except SyntaxError:
            continue
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef) and len(node.args.args) == 2 and node.args.args[1].arg == "fieldset" \
                    and node.args.args[0].arg in ("particles", "particle", "p"):
                node.decorator_list = []
                yield node.name, ast.unparse(node)


class _Grid:
    is_curvilinear = False


class _Field:
    grid = _Grid()


class _Vector:
    U = _Field()
    V = _Field()


class _FS:
    def __init__(self, fields, context):
        self.fields, self.context = fields, context


def _root(node):
    while isinstance(node, (ast.Subscript, ast.Attribute)):
        node = node.value
    return node.id if isinstance(node, ast.Name) else None


def permissive_context(src):
    """Everything the kernel mentions exists: `particles.<name>` (also through selections bound to locals) are float32 Variables,
    `fieldset.<name>[...]` fields on a rectilinear grid (vector fields when unpacked into a tuple), other `fieldset.<name>` constants."""
    import parcels_amd as pa

    tree = ast.parse(src)
    fdef = tree.body[0]
    pname, fname = fdef.args.args[0].arg, fdef.args.args[1].arg
    core = {v.name for v in pa.get_default_particle(np.float32).variables} | {"ei"}
    selections = {pname}
    for _ in range(3):
        for n in ast.walk(tree):
            if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name) and isinstance(n.value, ast.Subscript) \
                    and _root(n.value) in selections:
                selections.add(n.targets[0].id)
    variables, fields, consts = [], {}, {}
    sampled, vectors = set(), set()
    for n in ast.walk(tree):
        if isinstance(n, ast.Subscript) and isinstance(n.value, ast.Attribute) and isinstance(n.value.value, ast.Name) and n.value.value.id == fname:
            sampled.add(n.value.attr)
        if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Tuple) and isinstance(n.value, ast.Subscript) and isinstance(n.value.value, ast.Attribute) \
                and _root(n.value.value) == fname:
            vectors.add(n.value.value.attr)
    for n in ast.walk(tree):
        if isinstance(n, ast.Attribute):
            if _root(n.value) in selections and not (isinstance(n.value, ast.Name) and n.value.id == fname) and n.attr not in core \
                    and n.attr not in variables and not n.attr.startswith("_") and n.attr not in ("shape", "flatten", "size"):
                variables.append(n.attr)
            if isinstance(n.value, ast.Name) and n.value.id == fname:
                if n.attr in sampled:
                    fields[n.attr] = _Vector() if (n.attr in ("UV", "UVW") or n.attr in vectors) else _Field()
                else:
                    consts[n.attr] = 1.0
    pclass = pa.get_default_particle(np.float32).add_variable(
        [pa.Variable(v, dtype=np.float64 if v == "next_dt" else np.float32, initial=0) for v in variables])
    var_slot = {v: (i, "f64" if v == "next_dt" else "f32") for i, v in enumerate(variables) if v != "next_dt"}
    field_ids = {name: i for i, name in enumerate(fields)}
    return pclass, _FS(fields, consts), var_slot, field_ids


def free_names(src):
    """Module-level names the kernel reads that the snippet does not define (constants of the tutorial's notebook): given a value here."""
    import builtins

    tree = ast.parse(src)
    fdef = tree.body[0]
    bound = {a.arg for a in fdef.args.args} | {"np", "math", "parcels", "StatusCode"} | set(dir(builtins))
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
            bound.add(n.id)
        if isinstance(n, (ast.FunctionDef, ast.Lambda)) and n is not fdef:
            bound.add(getattr(n, "name", ""))
    out = {}
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound:
            out[n.id] = "[60, 61]" if n.id.isupper() and n.id.endswith("STATES") else "1.5"
    return out


def load_function(name, src, tmpdir, k):
    path = os.path.join(tmpdir, f"k{k}.py")
    with open(path, "w") as f:
        f.write("import math\nimport numpy as np\nimport parcels_amd as parcels\nfrom parcels_amd import StatusCode\n\n"
                + "".join(f"{k} = {v}\n" for k, v in free_names(src).items()) + "\n" + src + "\n")
    spec = importlib.util.spec_from_file_location(f"_survey_k{k}", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return getattr(mod, name)


def survey():
    """-> [(file, kernel name, 'translated' | 'host path' | 'error', reason)] over the distinct kernels of the reference's tests and docs."""
    from parcels_amd import jit

    files = sorted(glob.glob(os.path.join(REF, "tests", "**", "*.py"), recursive=True)
                   + glob.glob(os.path.join(REF, "docs", "**", "*.ipynb"), recursive=True)
                   + glob.glob(os.path.join(REF, "docs", "**", "*.md"), recursive=True))
    seen, rows = set(), []
    with tempfile.TemporaryDirectory() as tmp:
        for path in files:
            for name, src in kernel_defs(path):
                if src in seen:
                    continue
                seen.add(src)
                rel = os.path.relpath(path, REF)
                try:
                    func = load_function(name, src, tmp, len(rows))
                    pclass, fs, var_slot, field_ids = permissive_context(textwrap.dedent(src))
                    jit.translate(func, pclass, fs, var_slot, field_ids)
                    rows.append((rel, name, "translated", ""))
                except jit.NotTranslatable as e:
                    rows.append((rel, name, "host path", str(e)))
                except Exception as e:  # noqa: BLE001 -- a survey: report, do not stop
                    rows.append((rel, name, "error", f"{type(e).__name__}: {e}"))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--json")
    a = ap.parse_args()
    rows = survey()
    tally = collections.Counter(r[2] for r in rows)
    reasons = collections.Counter(r[3] for r in rows if r[2] != "translated")
    print(f"{len(rows)} distinct kernels: " + ", ".join(f"{k} {v}" for k, v in tally.items()))
    for reason, n in reasons.most_common():
        print(f"  {n:3d}  {reason}")
    if a.verbose:
        for r in rows:
            print(f"{r[2]:11s} {r[0]}::{r[1]}  {r[3]}")
    if a.json:
        json.dump({"kernels": len(rows), "tally": tally, "reasons": reasons, "rows": rows}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
