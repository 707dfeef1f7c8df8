#!/usr/bin/env python3
"""Poor man's pyflakes (none in the image): names loaded but never bound in a module, and `self.x` read in a class that
never stores `x` (nor inherits: classes with bases are skipped).  usage: tools/check_names.py files..."""
import ast
import builtins
import sys


def check(path):
    tree = ast.parse(open(path).read())
    defined = set(dir(builtins)) | {"__file__", "__name__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            defined.add(n.name)
        elif isinstance(n, ast.Import):
            defined.update((a.asname or a.name).split(".")[0] for a in n.names)
        elif isinstance(n, ast.ImportFrom):
            defined.update(a.asname or a.name for a in n.names)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            defined.add(n.id)
        elif isinstance(n, ast.arg):
            defined.add(n.arg)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            defined.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            defined.update(n.names)
    return sorted({n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in defined})


def check_self_attrs(path):
    out = []
    for cls in [n for n in ast.walk(ast.parse(open(path).read())) if isinstance(n, ast.ClassDef) and not n.bases]:
        stored = {n.name for n in cls.body if isinstance(n, (ast.FunctionDef, ast.ClassDef))}
        loaded = set()
        for n in ast.walk(cls):
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
                stored.add(n.id)
            if isinstance(n, ast.Constant) and isinstance(n.value, str):
                stored.add(n.value)  # __slots__ entries, setattr(self, "x", ...)
            if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == "self":
                (stored if isinstance(n.ctx, (ast.Store, ast.Del)) else loaded).add(n.attr)
        out += ["%s.%s" % (cls.name, a) for a in sorted(loaded - stored) if not a.startswith("__")]
    return out


bad = 0
for f in sys.argv[1:]:
    m = check(f) + check_self_attrs(f)
    if m:
        bad += 1
        print(f, "undefined:", m)
sys.exit(1 if bad else 0)
