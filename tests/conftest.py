import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def pytest_collection_modifyitems(config, items):
    """A test that hangs (a deadlocked worker pool, a kernel that never returns) must fail, not stall the round: every test gets a per-test limit where the
    pytest-timeout plug-in is installed (it is in this image), unless the command line or the test set one."""
    if not config.pluginmanager.hasplugin("timeout") or config.getoption("timeout", None):
        return
    for it in items:
        if it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(900 if it.get_closest_marker("gpu") is None else 600))


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as g
    g.build()
    return True


def pytest_sessionfinish(session, exitstatus):
    """Memory-safety runs (tools/asan_run.sh: LMPC_GUARD_REPORT=1, library built with -DLMPC_GUARD): report how many guard zones behind device buffers
    were found overwritten when the buffers were freed; any overrun fails the session."""
    if os.environ.get("LMPC_GUARD_REPORT") != "1":
        return
    import gc
    gc.collect()                                       # contexts still alive free (and check) their buffers now
    from racinglmpc_amd import _capi
    if _capi._lib is not None:
        n = int(_capi._lib.lmpc_debug_guard_failures())
        print("\nliblmpc_hip guard zones overwritten: %d" % n)
        if n:
            session.exitstatus = 1
