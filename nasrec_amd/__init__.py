"""nasrec_amd — MI355X-native engine for the NASRec supernet forward/backward/optimizer hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every kernel is hand-written
HIP for gfx950 behind the C-ABI in include/nasrec_hip.h (nasrec_amd/_lib.py is the ctypes binding).
"""
__version__ = "0.1.0"


# The plan compiler, the engine's host side and the level scheduler also exist as compiled extension modules (the same .py files
# through Cython, built in-tree by `__graft_entry__.build()` into nasrec_amd/_hostbuild/: a sampled supernet path is compiled on the host
# every step, and the compiled walk takes ~40 % less time).  Which of the two runs is decided HERE, by content: a finder in front of the
# default ones hands out `_hostbuild/<mod>.*.so` for `nasrec_amd.<mod>` only when the SHA-256 of `<mod>.py` as it is now equals the one
# recorded at build time (`<mod>.sha256`, and again inside the extension as `__source_sha256__`); otherwise — no build, a build of
# other sources, NASREC_NO_CYTHON=1 — the default finder imports the .py.  Nothing in the tree is renamed or removed at import (an
# rsync / checkout / tar that reorders mtimes changes nothing, a read-only install works), and one run with NASREC_NO_CYTHON=1 leaves
# later runs alone.
import importlib.abc
import importlib.machinery
import importlib.util
import os as _os
import sys as _sys

HOST_EXT_MODULES = ("plan", "engine", "schedule")
HOSTBUILD_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_hostbuild")


def source_sha256(mod: str) -> str:
    import hashlib
    with open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), mod + ".py"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def compiled_host_module(mod: str):
    """path of the compiled build of nasrec_amd.<mod> that matches the current source, or None"""
    if _os.environ.get("NASREC_NO_CYTHON") == "1" or mod not in HOST_EXT_MODULES:
        return None
    try:
        with open(_os.path.join(HOSTBUILD_DIR, mod + ".sha256")) as f:
            recorded = f.read().strip()
        if recorded != source_sha256(mod):
            return None
        for suffix in importlib.machinery.EXTENSION_SUFFIXES:
            so = _os.path.join(HOSTBUILD_DIR, mod + suffix)
            if _os.path.exists(so):
                return so
    except OSError:
        pass
    return None


class _HostBuildFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("nasrec_amd.") or fullname.count(".") != 1:
            return None
        so = compiled_host_module(fullname.split(".")[1])
        if so is None:
            return None
        return importlib.util.spec_from_file_location(fullname, so, loader=importlib.machinery.ExtensionFileLoader(fullname, so))


if not any(isinstance(f, _HostBuildFinder) for f in _sys.meta_path):
    _sys.meta_path.insert(0, _HostBuildFinder())


def host_modules_compiled():
    """{module: True if the compiled build is the one imported} — for reports and tests"""
    out = {}
    for m in HOST_EXT_MODULES:
        mod = _sys.modules.get("nasrec_amd." + m)
        out[m] = bool(mod is not None and getattr(mod, "__file__", "").endswith(tuple(importlib.machinery.EXTENSION_SUFFIXES)))
    return out
