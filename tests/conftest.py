import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


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
