"""The GPU paths cannot run in the authoring container: at least no module may use a name it never binds
(a poor man's pyflakes over the package, the bench and the tools)."""
import ast
import builtins
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _undefined(path):
    tree = ast.parse(open(path).read())
    bound = set(dir(builtins)) | {"__file__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            bound.add(n.name)
            if not isinstance(n, ast.ClassDef):
                for a in n.args.args + n.args.kwonlyargs + n.args.posonlyargs:
                    bound.add(a.arg)
                for a in (n.args.vararg, n.args.kwarg):
                    if a:
                        bound.add(a.arg)
        elif isinstance(n, ast.Lambda):
            for a in n.args.args + n.args.kwonlyargs:
                bound.add(a.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                bound.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            bound.add(n.id)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
        elif isinstance(n, ast.Global):
            bound.update(n.names)
    return [(n.lineno, n.id) for n in ast.walk(tree)
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound]


def test_no_module_uses_a_name_it_never_binds():
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for pat in ("mrgcn_amd/**/*.py", "tools/**/*.py", "oracle/**/*.py", "tests/*.py"):
        files += glob.glob(os.path.join(ROOT, pat), recursive=True)
    bad = {os.path.relpath(f, ROOT): u for f in files if (u := _undefined(f))}
    assert not bad, bad
