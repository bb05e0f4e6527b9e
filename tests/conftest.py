import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size CPU oracle cases (tens of seconds)")
    config.addinivalue_line("markers", "oracle_launch: device step that feeds a background CPU-oracle job (ordered first)")
    config.addinivalue_line("markers", "oracle_join: asserts on a background CPU-oracle job (ordered last)")
    config.addinivalue_line("markers", "oracle_jobs(kind): the test uses jobs of tests/oracle_pool.py that can start when collection ends")
    config.addinivalue_line("markers", "oracle_heavy: tens of seconds of IN-PROCESS oracle work: ordered behind the ordinary tests, when the pool's "
                                       "jobs (8 x 8 threads) have mostly finished and no longer compete for the host's memory bandwidth")


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(config, items):
    """The long CPU-oracle computations of the GPU suite run as background jobs (tests/oracle_pool.py).  Tests that produce what a
    job needs from the device run first, tests that wait for a job run last (stable otherwise), and the jobs that need nothing
    from the device are submitted right here - before the first test - for the cases that were actually selected."""
    rank = lambda it: 0 if it.get_closest_marker("oracle_launch") else (
        3 if it.get_closest_marker("oracle_join") else (2 if it.get_closest_marker("oracle_heavy") else 1))
    items.sort(key=rank)
    if any(it.get_closest_marker("gpu") for it in items):
        # in-process oracle calls (b <= 8 pairs per reference call) are FASTEST on ~32 threads of the GPU box's 256
        # (profiles/r04_oracle_threads.txt: 30 pairs/s on 32 threads, 11 on 128); the pool's jobs take 8 x 8 more
        import torch
        torch.set_num_threads(min(32, os.cpu_count() or 8))
    kinds = {}
    for it in items:
        m = it.get_closest_marker("oracle_jobs")
        if m is not None:
            kinds.setdefault(m.args[0], set()).add(it.callspec.params.get("name") if hasattr(it, "callspec") else None)
    if not kinds or config.option.collectonly:
        return
    if "sampled" in kinds:
        from tests import sampled_case
        sampled_case.prelaunch(sorted(n for n in kinds["sampled"] if n))
    if "trajectory" in kinds:
        from tests import trajectory_case
        trajectory_case.prelaunch(sorted(n for n in kinds["trajectory"] if n))


_T0 = None


def pytest_runtest_logreport(report):
    """``SGC_TEST_CLOCK=1``: wall-clock offset of every test's end (where does a slow suite spend its time: profiles/r04_gputest_final4.txt)."""
    global _T0
    if os.environ.get("SGC_TEST_CLOCK") != "1" or report.when != "call":
        return
    import time
    if _T0 is None:
        _T0 = time.time() - report.duration
    print("\n[clock] %7.1f s  %6.1f s  %s" % (time.time() - _T0, report.duration, report.nodeid), flush=True)


def pytest_sessionfinish(session, exitstatus):
    if "tests.oracle_pool" in sys.modules:
        sys.modules["tests.oracle_pool"].shutdown()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
