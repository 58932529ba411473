import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# contexts created by the tests keep the deterministic sample counters unless a test asks otherwise (the library's own
# default is the production kernels, flags = 0)
os.environ.setdefault("MOONRT_DEFAULT_FLAGS", "1")
# ... and send even the small test frames through the path queue (the library keeps the paths of launches below ~8 M samples
# inside the render wave: same result, one kernel instead of three)
os.environ.setdefault("MOONRT_PATH_QUEUE_MIN", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """libmoonrt.so, built in-tree if stale (hipcc cross-compiles without a GPU)."""
    from moonrtx_amd import build, _lib
    build.build_native()
    return _lib.load()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import orc
    return orc.lib()
