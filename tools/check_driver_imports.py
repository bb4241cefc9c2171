"""Does the import block of the UNCHANGED reference drivers resolve against shims/ ?

    python tools/check_driver_imports.py [/root/reference]

Parses the import statements of NeighborOverlap_large.py, NeighborOverlap_large_ppa.py and NeighborOverlapCitation2.py
(read as text; nothing of the reference is executed) and imports every one of them in a child process whose
PYTHONPATH is `shims:repo`.  Exit code 0 = every name resolves.  Runs in the build container (the reference never
travels to the GPU box)."""
import ast
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVERS = ("NeighborOverlap_large.py", "NeighborOverlap_large_ppa.py", "NeighborOverlapCitation2.py")


def import_lines(path):
    tree = ast.parse(open(path).read())
    out = []
    for node in tree.body:
        if isinstance(node, ast.Import):
            for a in node.names:
                out.append((node.lineno, f"import {a.name}"))
        elif isinstance(node, ast.ImportFrom) and node.level == 0:
            out.append((node.lineno, f"from {node.module} import " + ", ".join(a.name for a in node.names)))
    return out


def main(ref):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "shims"), ROOT]))
    bad = 0
    for d in DRIVERS:
        path = os.path.join(ref, d)
        if not os.path.exists(path):
            print(f"{d}: not found under {ref}")
            bad += 1
            continue
        lines = sorted(set(import_lines(path)))
        stmts = sorted({s for _, s in lines})
        prog = "import sys\nbad = 0\n" + "".join(
            f"try:\n    {s}\nexcept Exception as e:\n    bad += 1; print('FAIL', {s!r}, type(e).__name__, e)\n" for s in stmts) + "sys.exit(bad)\n"
        r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, cwd="/tmp")
        print(f"{d}: {len(stmts)} distinct import statements, {r.returncode} unresolved")
        if r.returncode:
            print(r.stdout + r.stderr[-2000:])
        bad += r.returncode
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference") else 0)
