"""CPU-oracle jobs that run BESIDE the GPU tests (TEST INFRASTRUCTURE).

The f32 oracle is slow (10-15 ordered pairs per second, forward + backward) and the host cores idle while the GPU tests run, so the
long oracle computations of ``pytest -m gpu`` - the sampled steps of the full-size minibatches (forward, backward, backward with
the device's routes) and the multi-step training trajectories - are submitted as jobs to a small process pool
(``tests/oracle_worker.py``: one process per job, inputs regenerated from seeds) as early as possible and collected by the tests
that assert on them.  Jobs that depend on nothing from the device are submitted when collection ends (``tests/conftest.py``).
A test that finds its job missing (run on its own with ``-k``) submits it and waits: the results do not depend on the schedule.
"""
import atexit
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_lock = threading.Lock()
_jobs = {}            # name -> dict(dir, spec, proc, state: queued | running | done)
_order = []           # submission order of queued jobs
_root = None


def _plan():
    """(concurrent jobs, torch threads per job).  One oracle call of b = 8 pairs does not scale with threads - measured on the GPU
    box's 256 host threads (tools/oracle_threads.py -> profiles/r04_oracle_threads.txt): 1 process x 8 / 32 / 128 threads = 20 / 30 /
    11 pairs/s, 8 processes x 8 threads = 52 pairs/s in total - so a many-core host runs several narrow jobs side by side."""
    n = os.cpu_count() or 8
    env = os.environ.get("SGC_ORACLE_POOL")              # "jobs x threads", e.g. 6x16
    if env:
        j, t = env.lower().split("x")
        return int(j), int(t)
    if n <= 16:
        return 1, n
    return min(8, n // 8), 8


def _tmp_root():
    global _root
    if _root is None:
        base = None
        for cand in ("/dev/shm", tempfile.gettempdir()):
            try:
                st = os.statvfs(cand)
                if st.f_bavail * st.f_frsize > 48 << 30:           # trajectories keep ~11 GB of updates each until they are read
                    base = cand
                    break
            except OSError:
                pass
        _root = tempfile.mkdtemp(prefix="sgc_oracle_", dir=base)
        atexit.register(shutdown)
    return _root


def _start(name):
    job = _jobs[name]
    env = dict(os.environ)
    env.update(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(job["spec"]["threads"]),
               PYTHONDONTWRITEBYTECODE="1")
    log = open(os.path.join(job["dir"], "log.txt"), "w")
    job["t0"] = time.time()
    job["proc"] = subprocess.Popen([sys.executable, "-m", "tests.oracle_worker", job["dir"]], cwd=REPO, env=env, stdout=log,
                                   stderr=subprocess.STDOUT)
    job["state"] = "running"


def _pump():
    """Start queued jobs while slots are free (called under the lock)."""
    slots, _ = _plan()
    running = 0
    for j in _jobs.values():
        if j["state"] == "running":
            if j["proc"].poll() is None:
                running += 1
            else:
                j["state"] = "done"
                j["t1"] = time.time()
    while _order and running < slots:
        _start(_order.pop(0))
        running += 1


def _watch():
    while True:
        time.sleep(0.5)
        with _lock:
            _pump()
            if not _order and all(j["state"] == "done" for j in _jobs.values()):
                return


_watcher = None


def submit(name, spec, front=False):
    """Queue a job (no-op if a job of that name exists).  ``spec``: see ``tests/oracle_worker.py``."""
    global _watcher
    with _lock:
        if name in _jobs:
            return
        d = os.path.join(_tmp_root(), name)
        os.makedirs(d)
        spec = dict(spec, threads=_plan()[1])
        torch.save(spec, os.path.join(d, "job.pt"))
        _jobs[name] = dict(dir=d, spec=spec, proc=None, state="queued")
        if front:
            _order.insert(0, name)
        else:
            _order.append(name)
        _pump()
        if _watcher is None or not _watcher.is_alive():
            _watcher = threading.Thread(target=_watch, daemon=True)
            _watcher.start()


def submitted(name):
    return name in _jobs


def job_dir(name):
    return _jobs[name]["dir"]


def _fail(name):
    job = _jobs[name]
    log = open(os.path.join(job["dir"], "log.txt")).read()[-3000:]
    raise RuntimeError("oracle job %s failed (exit %s):\n%s" % (name, job["proc"].returncode, log))


def wait_file(name, fname, timeout=3600.0):
    """Path of a file the job writes (atomically) as soon as it exists."""
    job = _jobs[name]
    path = os.path.join(job["dir"], fname)
    t0 = time.time()
    while not os.path.exists(path):
        with _lock:
            _pump()
        if job["state"] == "done" and not os.path.exists(path):
            _fail(name)
        if time.time() - t0 > timeout:
            raise TimeoutError("oracle job %s: %s not written within %.0f s" % (name, fname, timeout))
        time.sleep(0.2)
    return path


def result(name, timeout=3600.0):
    """The job's ``out.pt`` (waits for it)."""
    path = wait_file(name, "out.pt", timeout)
    out = torch.load(path)
    job = _jobs[name]
    print("[oracle pool] %s: %.0f s in the worker (%d threads), waited on from %.0f s after its start"
          % (name, (job.get("t1") or time.time()) - job["t0"], job["spec"]["threads"], time.time() - job["t0"]))
    return out


def release(name):
    """Delete a finished job's files (the big ones live in memory-backed storage when /dev/shm is used)."""
    job = _jobs.get(name)
    if job is not None and job["state"] == "done":
        shutil.rmtree(job["dir"], ignore_errors=True)


def shutdown():
    global _root
    with _lock:
        _order.clear()
        for j in _jobs.values():
            if j["proc"] is not None and j["proc"].poll() is None:
                j["proc"].kill()
        _jobs.clear()
    if _root is not None:
        shutil.rmtree(_root, ignore_errors=True)
        _root = None
