"""Crude undefined-name check (no pyflakes in this image): python tools/undefined_names.py file.py ...  -- names that are loaded
somewhere in a file but never bound anywhere in it (GPU tests cannot be executed in the build container; this catches the typo class
of failure before a GPU run is spent on it)."""
import ast
import builtins
import sys

bad = 0
for path in sys.argv[1:]:
    tree = ast.parse(open(path).read())
    defined = set(dir(builtins)) | {"__file__", "__name__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            defined.add(n.name)
        elif isinstance(n, ast.Import):
            defined.update((a.asname or a.name).split(".")[0] for a in n.names)
        elif isinstance(n, ast.ImportFrom):
            defined.update(a.asname or a.name for a in n.names)
        elif isinstance(n, ast.arg):
            defined.add(n.arg)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            defined.add(n.id)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            defined.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            defined.update(n.names)
    undef = sorted({n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in defined})
    if undef:
        bad += 1
        print(f"{path}: {undef}")
sys.exit(1 if bad else 0)
