"""nasrec_amd — MI355X-native engine for the NASRec supernet forward/backward/optimizer hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every kernel is hand-written
HIP for gfx950 behind the C-ABI in include/nasrec_hip.h (nasrec_amd/_lib.py is the ctypes binding).
"""
__version__ = "0.1.0"


# The plan compiler, the engine's host side and the level scheduler also exist as compiled extension modules (the same .py files
# through Cython, built in-tree by `__graft_entry__.build()`: a sampled supernet path is compiled on the host every step, and the
# compiled walk takes ~40 % less time).  Python imports `plan.cpython-*.so` ahead of `plan.py`; a build that is OLDER than its source
# would silently run old code, so stale builds are set aside here, before any submodule is imported (the .py then takes over).
def _drop_stale_host_builds():
    import glob
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    for mod in HOST_EXT_MODULES:
        src = os.path.join(here, mod + ".py")
        for so in glob.glob(os.path.join(here, mod + ".*.so")):
            try:
                if os.path.getmtime(so) < os.path.getmtime(src) or os.environ.get("NASREC_NO_CYTHON") == "1":
                    os.replace(so, so + ".stale")
            except OSError:
                pass


HOST_EXT_MODULES = ("plan", "engine", "schedule")
_drop_stale_host_builds()
