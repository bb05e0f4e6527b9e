"""ctypes binding of the C-ABI library (include/sgc_relhead.h).  Fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes
import os

_LIB = None


class HipExtensionMissing(RuntimeError):
    pass


def lib_path() -> str:
    from .build import lib_path as _p
    return _p()


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    """Load libsgc_relhead.so (building it in-tree with hipcc when absent)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch must be loaded first: it brings the HIP runtime (libamdhip64) this library's kernels have to register
    # with.  Loading the library before torch leaves two runtimes in the process and every launch fails.
    import torch  # noqa: F401
    path = lib_path()
    from .build import build, is_stale
    if is_stale():                      # missing, or compiled from other sources than the tree holds (content hash, not mtimes)
        if not build_if_missing:
            raise HipExtensionMissing(path + " is missing or stale; run `python -m scene_graph_commonsense_amd.build`")
        build(verbose=False)
    try:
        _LIB = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise HipExtensionMissing("cannot load %s: %s" % (path, e))
    return _LIB


def check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError("%s failed with status %d (1=bad argument, 2=launch error)" % (what, status))


def ptr(t):
    """Device (or host) pointer of a torch tensor / None as a void*."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
