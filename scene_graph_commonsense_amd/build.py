"""Build the C-ABI shared library (HIP kernels for gfx950) in-tree with hipcc.

    python -m scene_graph_commonsense_amd.build [--force]

Produces scene_graph_commonsense_amd/lib/libsgc_relhead.so.  hipcc cross-compiles for gfx950 without a
GPU, so this also runs in the CPU-only build container (``__graft_entry__.build()``).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIBNAME = "libsgc_relhead.so"
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
         "-I", CSRC, "-I", os.path.join(os.path.dirname(HERE), "include")]
if os.environ.get("SGC_EXPERIMENTS") == "1":      # also compile the rejected main-loop variants (tools/gemm_microbench.py)
    FLAGS.insert(5, "-DSGC_EXPERIMENTS")


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_dep() -> float:
    t = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            t = max(t, os.path.getmtime(os.path.join(root, f)))
    return t


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def source_hash() -> str:
    """Content hash of everything the library is compiled from (csrc/ + include/ + the flags).  The staleness test uses it
    rather than mtimes: a snapshot copied to another machine keeps contents, not timestamps."""
    import hashlib
    h = hashlib.sha1(" ".join(f for f in FLAGS if f.startswith("-") and f != "-I").encode())
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in sorted(os.listdir(root)):
            h.update(f.encode())
            with open(os.path.join(root, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def _stamp_path() -> str:
    return os.path.join(LIBDIR, "source_hash.txt")


def is_stale() -> bool:
    """True when the built library is missing or was compiled from other sources than the ones in the tree."""
    if not os.path.exists(lib_path()) or not os.path.exists(_stamp_path()):
        return True
    with open(_stamp_path()) as f:
        return f.read().strip() != source_hash()


def _compile(src: str) -> str:
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) >= _newest_dep():
        return obj
    cmd = [hipcc()] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    out = lib_path()
    if not force and not is_stale():
        return out
    if force or not os.path.exists(_stamp_path()):
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(_stamp_path(), "w") as f:
        f.write(source_hash())
    if verbose:
        print("built", out)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
