"""Collects, by parsing (ast) the reference's own sources in the build container, every name the reference imports from a
module that bmhrl_amd.install aliases -> tests/golden/boundary_names.json.  tests/test_boundary_cpu.py asserts that each of
them resolves after `import bmhrl_amd.install` (the drop-in boundary).  Only names are recorded."""
import ast
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ALIASES = None


def aliased_modules():
    src = open(os.path.join(ROOT, "bmhrl_amd", "install.py")).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "ALIASES":
            return sorted(ast.literal_eval(node.value))
    raise SystemExit("ALIASES not found")


def main():
    mods = set(aliased_modules())
    out = {}
    for dirpath, _, files in os.walk(REF):
        for f in sorted(files):
            if not f.endswith(".py"):
                continue
            path = os.path.join(dirpath, f)
            rel = os.path.relpath(path, REF)
            try:
                tree = ast.parse(open(path, encoding="utf-8", errors="replace").read())
            except SyntaxError:
                continue
            pkg = os.path.dirname(rel).replace(os.sep, ".")
            for node in ast.walk(tree):
                if not isinstance(node, ast.ImportFrom):
                    continue
                mod = node.module or ""
                if node.level:                                    # relative import: resolve against the file's package
                    base = pkg.split(".") if pkg else []
                    base = base[:len(base) - (node.level - 1)] if node.level > 1 else base
                    mod = ".".join(base + ([mod] if mod else []))
                if mod in mods:
                    for a in node.names:
                        out.setdefault(mod, {}).setdefault(a.name, []).append(f"{rel}:{node.lineno}")
    res = {m: {n: sorted(set(w)) for n, w in sorted(names.items())} for m, names in sorted(out.items())}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "boundary_names.json")
    json.dump(res, open(dst, "w"), indent=1)
    print(dst, {m: len(v) for m, v in res.items()})


if __name__ == "__main__":
    main()
