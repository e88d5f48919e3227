"""Import the REAL reference (/root/reference) for the fixture generators — and make it impossible to pick up this
repository's own `nasrec/` import-path shim instead.

`/root/reference/nasrec` has no `__init__.py` (a namespace package); a regular package of the same name anywhere on
`sys.path` wins over it regardless of path order.  So the generators (a) never put the repository root on `sys.path`,
(b) strip any entry that holds a regular `nasrec` package, (c) assert after the import that every `nasrec.*` module
came from /root/reference.  The oracle (needed for the name-seeded weights) is loaded by FILE PATH.
"""
import importlib.util
import os
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def setup():
    sys.dont_write_bytecode = True
    if not os.path.isdir(os.path.join(REF, "nasrec")):
        raise SystemExit("the reference is not mounted at %s: fixtures can only be regenerated in the build container" % REF)
    for m in [m for m in sys.modules if m == "nasrec" or m.startswith("nasrec.") or m == "nasrec_amd" or m.startswith("nasrec_amd.")]:
        del sys.modules[m]
    clean = []
    for p in sys.path:
        q = os.path.abspath(p or os.getcwd())
        if q == ROOT or os.path.isfile(os.path.join(q, "nasrec", "__init__.py")):
            continue
        clean.append(p)
    sys.path[:] = [REF] + [p for p in clean if os.path.abspath(p or ".") != REF]


def load_oracle():
    """oracle/nasrec_oracle.py by file path (the repository root stays off sys.path)"""
    spec = importlib.util.spec_from_file_location("nasrec_oracle_for_golden", os.path.join(ROOT, "oracle", "nasrec_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def assert_reference_modules():
    bad = []
    for name, mod in list(sys.modules.items()):
        if name == "nasrec" or name.startswith("nasrec."):
            f = getattr(mod, "__file__", None)
            paths = [f] if f else list(getattr(mod, "__path__", []))
            if not paths or not all(os.path.abspath(p).startswith(REF + os.sep) for p in paths):
                bad.append((name, paths))
    assert not bad, "these modules did not come from the reference: %s" % bad
    assert "nasrec_amd" not in sys.modules, "the build's own package was imported into a fixture generator"
