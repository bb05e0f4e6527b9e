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


def _headers_hash() -> str:
    """Hash of the flags and of every header a translation unit may include (csrc/*.h, include/*.h)."""
    import hashlib
    h = hashlib.sha1(" ".join(f for f in FLAGS if f.startswith("-") and f != "-I").encode())
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in sorted(os.listdir(root)):
            if f.endswith(".h"):
                h.update(f.encode())
                with open(os.path.join(root, f), "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()


def _compile(src: str, hdr_hash: str) -> str:
    """One object per (source content, headers, flags): the object's name carries that hash, so a changed flag (``SGC_EXPERIMENTS``)
    or header can never be satisfied by an object compiled under another one; older objects of the same source are removed."""
    import hashlib
    base = os.path.basename(src)[:-4]
    with open(src, "rb") as fh:
        key = hashlib.sha1(hdr_hash.encode() + fh.read()).hexdigest()[:12]
    obj = os.path.join(OBJDIR, "%s.%s.o" % (base, key))
    for f in os.listdir(OBJDIR):
        if f.startswith(base + ".") and f.endswith(".o") and f != os.path.basename(obj):
            os.remove(os.path.join(OBJDIR, f))
    if os.path.exists(obj):
        return obj
    tmp = obj + ".tmp%d" % os.getpid()
    cmd = [hipcc()] + FLAGS + ["-c", src, "-o", tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    os.replace(tmp, obj)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile and link under an exclusive file lock (several ranks may find the library stale at once: one builds, the others
    wait and then see a fresh stamp); the library and its stamp are moved into place atomically."""
    import fcntl
    os.makedirs(OBJDIR, exist_ok=True)
    out = lib_path()
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():
                return out
            if force:
                for f in os.listdir(OBJDIR):
                    os.remove(os.path.join(OBJDIR, f))
            srcs = sources()
            hh = _headers_hash()
            with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
                objs = list(ex.map(lambda s_: _compile(s_, hh), srcs))
            tmp = out + ".tmp%d" % os.getpid()
            cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
            os.replace(tmp, out)
            with open(_stamp_path() + ".tmp", "w") as f:
                f.write(source_hash())
            os.replace(_stamp_path() + ".tmp", _stamp_path())
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if verbose:
        print("built", out)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
