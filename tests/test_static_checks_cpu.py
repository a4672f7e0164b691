"""Static checks of the host layer that need no GPU: a function-local `import x` makes `x` a local name for the WHOLE function,
so a use of the module-level `x` above that line raises UnboundLocalError at run time -- in code paths only a GPU box executes
(round 5: `tune_dense_gemms` read `os.environ` above a later `import hashlib, os`; the bench's prefill instance died at
start-up, found by the GPU suite, not here)."""
import ast
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _local_import_shadows(path):
    tree = ast.parse(open(path).read(), path)
    bad = []
    for fn in ast.walk(tree):
        if not isinstance(fn, (ast.FunctionDef, ast.AsyncFunctionDef)):
            continue
        # names bound by import statements directly in this function (not in nested functions)
        nested = {id(n) for sub in ast.walk(fn) if sub is not fn and isinstance(sub, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda))
                  for n in ast.walk(sub)}
        imported = {}
        for n in ast.walk(fn):
            if id(n) in nested or not isinstance(n, (ast.Import, ast.ImportFrom)):
                continue
            for a in n.names:
                name = (a.asname or a.name).split(".")[0]
                imported[name] = min(imported.get(name, n.lineno), n.lineno)
        for n in ast.walk(fn):
            if id(n) in nested or not isinstance(n, ast.Name) or not isinstance(n.ctx, ast.Load):
                continue
            if n.id in imported and n.lineno < imported[n.id]:
                bad.append(f"{os.path.relpath(path, ROOT)}:{n.lineno}: `{n.id}` is used in {fn.name}() above its local import "
                           f"(line {imported[n.id]})")
    return bad


def test_no_name_is_used_above_its_function_local_import():
    files = glob.glob(os.path.join(ROOT, "semi-pd_amd", "**", "*.py"), recursive=True) + [os.path.join(ROOT, "bench.py")]
    bad = [b for f in files for b in _local_import_shadows(f)]
    assert not bad, "\n".join(bad)
