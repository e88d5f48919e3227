"""Which build of the host modules (plan compiler, engine, level scheduler) a process imports: the compiled one
(nasrec_amd/_hostbuild/, the same sources through Cython) only while the SHA-256 recorded at build time equals the current source's —
decided by content in nasrec_amd/__init__.py, never by file times, and without touching the tree."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nasrec_amd")
BUILD = os.path.join(PKG, "_hostbuild")
PROBE = ("import json, nasrec_amd; from nasrec_amd import plan, engine, schedule; "
         "print(json.dumps([nasrec_amd.host_modules_compiled(), getattr(plan, '__source_sha256__', None), nasrec_amd.source_sha256('plan')]))")


def probe(env=None, cwd=ROOT):
    import json
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, "-c", PROBE], cwd=cwd, env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def have_build():
    return all(os.path.exists(os.path.join(BUILD, m + ".sha256")) for m in ("plan", "engine", "schedule"))


def test_compiled_host_modules_are_chosen_by_source_hash_not_by_file_time(tmp_path):
    if not have_build():
        pytest.skip("no compiled host modules in this tree (build() without Cython): the .py modules are what runs")
    used, embedded, now = probe()
    assert used == {"plan": True, "engine": True, "schedule": True}, used
    assert embedded == now, "the compiled plan module was built from other sources than nasrec_amd/plan.py"
    # NASREC_NO_CYTHON=1: this run takes the .py; the tree is left alone, so the next run is compiled again
    before = sorted(os.listdir(BUILD))
    used, embedded, _ = probe({"NASREC_NO_CYTHON": "1"})
    assert used == {"plan": False, "engine": False, "schedule": False} and embedded is None
    assert sorted(os.listdir(BUILD)) == before and not [f for f in os.listdir(PKG) if f.endswith(".stale")]
    assert probe()[0]["plan"] is True
    # a copy of the tree whose plan.py differs from what was compiled (older file time on the source, newer on the build: what an
    # rsync or a checkout can leave behind) runs the .py for that module and the compiled builds for the others
    dst = tmp_path / "tree"
    shutil.copytree(PKG, dst / "nasrec_amd", ignore=shutil.ignore_patterns("__pycache__", "lib"))
    os.symlink(os.path.join(PKG, "lib"), dst / "nasrec_amd" / "lib")
    src = dst / "nasrec_amd" / "plan.py"
    src.write_text(src.read_text() + "\n# edited after the build\n")
    os.utime(src, (1, 1))
    used, embedded, now = probe(cwd=str(dst), env={"PYTHONPATH": str(dst)})
    assert used == {"plan": False, "engine": True, "schedule": True}, used
    assert embedded is None
